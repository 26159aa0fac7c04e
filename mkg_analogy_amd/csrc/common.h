// Shared device helpers for the MarT/MKGformer gfx950 kernels.
// gfx950 (CDNA4) only: wave64, MFMA 32x32x16 bf16, LDS-DMA (global_load_lds), ds_read_b64_tr_b16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
// fp16: the FORWARD operands of the text stream (engine.text_f16).  Post-LayerNorm activations, GELU outputs and N(0, 0.02)-scale weights sit
// well inside fp16's range, and its 11-bit significand rounds 8x finer than bf16 at the same MFMA rate; gradients stay bf16 (range).
typedef _Float16 h16;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 h16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 h16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// ---- error plumbing shared by every translation unit (defined in util.hip)
extern "C" void mart_set_error(const char* msg);
#define MART_CHECK(cond, msg)            \
  do {                                   \
    if (!(cond)) {                       \
      mart_set_error(msg);               \
      return -1;                         \
    }                                    \
  } while (0)
#define MART_LAUNCH_CHECK()                               \
  do {                                                    \
    hipError_t e_ = hipGetLastError();                    \
    if (e_ != hipSuccess) {                               \
      mart_set_error(hipGetErrorString(e_));              \
      return -2;                                          \
    }                                                     \
  } while (0)

// ---- per-DEVICE "kernel attribute already set" flags (hipFuncSetAttribute is per device): a process that drives several GPUs
// sets the dynamic-LDS attribute once on each of them.  Two host threads racing on a first call both set it: harmless.
struct MartAttrOnce {
  bool done[64] = {};
  bool* slot() { int dev = 0; (void)hipGetDevice(&dev); return &done[dev & 63]; }
};

// ---- MFMA 32x32x16 bf16.  D[i][j] += sum_k A[i][k] B[k][j].
// Operand layout (wave64): lane l holds A[i = l&31][k = 8*(l>>5) + 0..7] and B[k = 8*(l>>5)+0..7][j = l&31].
// Result layout: lane l, reg r holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// the same product on fp16 operands (raw 16-bit lanes travel through LDS as bf16x8; only the matrix instruction differs)
__device__ __forceinline__ f32x16 mfma32h(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}
// row index inside a 32x32 MFMA result tile held by (lane-half h, reg r)
__device__ __forceinline__ int mfma_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- LDS transpose read: within each 16-lane group, lane p supplies the address of 4 contiguous bf16
// (row p>>2, 4-column chunk p&3 of a 4x16 block) and receives column p, rows 0..3 of that block.
__device__ __forceinline__ s16x4 lds_tr_read(const void* lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_addr));
}
__device__ __forceinline__ bf16x8 join_tr(s16x4 lo, s16x4 hi) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// ---- async 16-byte global -> LDS copy.  LDS destination = wave-uniform base + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc_lane, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc_lane), LDS_PTR(lds_wave_base), 16, 0, 0);
}

// Same copy issued as opaque assembly.  With the builtin the compiler inserts an s_waitcnt vmcnt(0) in front of the next
// ds_read_b64_tr_b16 (it cannot prove which LDS bytes the DMA writes), i.e. right after the prefetch of the NEXT tile has
// been issued -- which serialises the prefetch with the transposed fragment reads of the CURRENT tile.  Kernels that use
// this variant order the DMA themselves: an explicit s_waitcnt vmcnt(0) before the barrier that publishes the tile.
__device__ __forceinline__ void glds16_raw(const void* gsrc_lane, void* lds_wave_base) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(__UINTPTR_TYPE__)LDS_PTR(lds_wave_base));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(gsrc_lane) : "memory", "m0");
}

// ---- bf16 <-> f32
__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }
__device__ __forceinline__ bf16x4 f4_to_bf4(f32x4 v) {
  bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  return o;
}
// four floats -> four bf16 through two 2-element conversions (v_cvt_pk_bf16_f32 each, no repacking)
__device__ __forceinline__ bf16x4 f2x2_to_bf4(f32x2 lo, f32x2 hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
  const bf16x2_t l = __builtin_convertvector(lo, bf16x2_t), h = __builtin_convertvector(hi, bf16x2_t);
  const u32x2_t w = {__builtin_bit_cast(unsigned, l), __builtin_bit_cast(unsigned, h)};
  return __builtin_bit_cast(bf16x4, w);
}
// f32 -> fp16 with the 16-bit lanes typed as bf16x4 (the staging / store code moves raw 16-bit lanes)
__device__ __forceinline__ bf16x4 f4_to_h4raw(f32x4 v) { return __builtin_bit_cast(bf16x4, __builtin_convertvector(v, h16x4)); }
__device__ __forceinline__ bf16x4 f2x2_to_h4raw(f32x2 lo, f32x2 hi) { return f4_to_h4raw(f32x4{lo[0], lo[1], hi[0], hi[1]}); }
__device__ __forceinline__ f32x4 bf4_to_f4(bf16x4 v) {
  f32x4 o = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  return o;
}

// ---- activations (reference: transformers ACT2FN 'gelu' = erf GELU, 'quick_gelu' = x*sigmoid(1.702x))
#define ACT_NONE 0
#define ACT_GELU 1
#define ACT_QGELU 2
#define ACT_STORED 3   // act_grad only: the operand already is act'(z)
// sigmoid(1.702 z) with the hardware exp2 / rcp (1 ulp each): the IEEE division the plain expression compiles to is ten
// VALU instructions per element -- 128 elements per lane in a 256x256 GEMM epilogue, i.e. microseconds per tile.
__device__ __forceinline__ float qgelu_sigmoid(float z) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * (-1.702f * 1.44269504088896340736f)));
}
__device__ __forceinline__ float act_fwd(float z, int act) {
  if (act == ACT_GELU) return 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f));
  if (act == ACT_QGELU) return z * qgelu_sigmoid(z);
  return z;
}
// a = act(z), g = act'(z) from one evaluation of the sigmoid / erf
__device__ __forceinline__ void act_fwd_grad(float z, int act, float& a, float& g) {
  if (act == ACT_GELU) {
    const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * z * z);
    a = z * cdf; g = cdf + z * pdf;
  } else if (act == ACT_QGELU) {
    const float s = qgelu_sigmoid(z);
    a = z * s; g = s * (1.0f + 1.702f * z * (1.0f - s));
  } else { a = z; g = 1.0f; }
}
// two elements at a time: everything but the two transcendentals per element maps to packed f32 instructions (the GEMM
// epilogues are VALU-bound: one wave64 VALU instruction = 4 cycles of the SIMD, 128 elements per lane and tile)
__device__ __forceinline__ void act_fwd_grad2(f32x2 z, int act, f32x2& a, f32x2& g) {
  if (act == ACT_QGELU) {
    const f32x2 x = z * f32x2{-1.702f * 1.44269504088896340736f, -1.702f * 1.44269504088896340736f};
    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])} + f32x2{1.0f, 1.0f};
    const f32x2 s = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    a = z * s;
    g = s * ((z * f32x2{1.702f, 1.702f}) * (f32x2{1.0f, 1.0f} - s) + f32x2{1.0f, 1.0f});
  } else if (act == ACT_GELU) {
    // erf through Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 rounding of the two outputs this
    // feeds): erf(|u|) = 1 - t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-u^2), t = 1 / (1 + p |u|), u = z / sqrt(2).
    // exp(-u^2) = exp(-z^2 / 2) is the Gaussian of the derivative as well: ONE exp2 and ONE rcp per element instead of
    // erff() + expf() (~70 VALU instructions per element: 30 us per 256x256 tile, a third of the text FFN launches).
    const f32x2 one = {1.0f, 1.0f};
    const f32x2 az = {__builtin_fabsf(z[0]), __builtin_fabsf(z[1])};
    const f32x2 x = (z * z) * f32x2{-0.5f * 1.44269504088896340736f, -0.5f * 1.44269504088896340736f};
    const f32x2 e = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};          // exp(-z^2 / 2)
    const f32x2 d = az * f32x2{0.3275911f * 0.70710678118654752440f, 0.3275911f * 0.70710678118654752440f} + one;
    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2 pl = t * f32x2{1.061405429f, 1.061405429f} + f32x2{-1.453152027f, -1.453152027f};
    pl = pl * t + f32x2{1.421413741f, 1.421413741f};
    pl = pl * t + f32x2{-0.284496736f, -0.284496736f};
    pl = pl * t + f32x2{0.254829592f, 0.254829592f};
    const f32x2 erfa = one - (pl * t) * e;                                                    // erf(|u|) in [0, 1]
    const f32x2 erfs = {__builtin_copysignf(erfa[0], z[0]), __builtin_copysignf(erfa[1], z[1])};
    const f32x2 cdf = erfs * f32x2{0.5f, 0.5f} + f32x2{0.5f, 0.5f};
    a = z * cdf;
    g = cdf + z * (e * f32x2{0.39894228040143267794f, 0.39894228040143267794f});
  } else {
    a = z; g = f32x2{1.0f, 1.0f};
  }
}
__device__ __forceinline__ float act_grad(float z, int act) {
  if (act == ACT_STORED) return z;
  if (act == ACT_GELU) {
    float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
    float pdf = 0.39894228040143267794f * __expf(-0.5f * z * z);
    return cdf + z * pdf;
  }
  if (act == ACT_QGELU) {
    const float s = qgelu_sigmoid(z);
    return s * (1.0f + 1.702f * z * (1.0f - s));
  }
  return 1.0f;
}

// ---- wave / block reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- counter-based RNG for dropout: one 32-bit hash per element index (stateless, so the backward
// pass regenerates the forward mask from (seed, index)).  keep <=> u >= p.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// Round 3: ONE hash per PAIR of element indices, 16 bits each (the three-round 64-bit form cost ~35 VALU instructions per element and made
// the 64 x 64 text attention kernels -- 64 dropout decisions per lane -- VALU-bound on the hash): element idx takes the low half of
// mix32((idx >> 1) ^ s2) when idx is even and the high half when odd; s2 folds the 64-bit seed and the upper index bits.  keep <=> u16 >= thr,
// thr = round(p * 65536) (p = 0.1 -> 0.100006).
__device__ __forceinline__ uint32_t rng_seedmix(uint64_t seed, uint32_t hi) { return mix32((uint32_t)(seed >> 32) + hi * 0x9e3779b9U) ^ (uint32_t)seed; }
__device__ __forceinline__ uint32_t dropout_thr16(float p) { return (uint32_t)(p * 65536.0f + 0.5f); }
__device__ __forceinline__ uint32_t rng_pair(uint32_t s2, uint32_t pair) { return mix32(pair ^ s2); }
// 32-bit index form (the caller guarantees idx < 2^32 and passes s2 = rng_seedmix(seed, 0), thr = dropout_thr16(p)): identical decisions
__device__ __forceinline__ bool dropout_keep32(uint32_t s2, uint32_t idx, uint32_t thr) {
  const uint32_t h = rng_pair(s2, idx >> 1);
  return ((idx & 1u) ? (h >> 16) : (h & 0xffffU)) >= thr;
}
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t idx, float p) {
  const uint64_t pair = idx >> 1;
  const uint32_t h = rng_pair(rng_seedmix(seed, (uint32_t)(pair >> 32)), (uint32_t)pair);
  return ((idx & 1u) ? (h >> 16) : (h & 0xffffU)) >= dropout_thr16(p);
}

// ---- XCD-aware, bijective remap of a linear workgroup id (8 XCDs; block b runs on XCD b%8):
// consecutive logical ids land on the same XCD so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int NX = 8;
  int xcd = bid % NX, idx = bid / NX;
  int q = nwg / NX, r = nwg % NX;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

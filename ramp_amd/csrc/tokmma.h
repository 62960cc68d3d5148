// Shared pieces of the token-owning kernels (ffx.hip, tkl.hip): vector types, the fp16x3 operand split, the delayed scale,
// Phi / phi for GEGLU, the 32x32x16 fp16 MFMA.
#pragma once
#include "core.h"

namespace ramp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* glb_ptr_t;
__device__ __forceinline__ void glds16(const void* src, char* dst) {
  __builtin_amdgcn_global_load_lds((glb_ptr_t)(uintptr_t)src, (lds_ptr_t)(unsigned)(uintptr_t)dst, 16, 0, 0);
}

__device__ __forceinline__ float scale_from(float mx) {           // 2^(5 - floor(log2 max)): the operand lands in [2^5, 2^6)
  float s = 1.f;
  if (mx > 0.f) {
    int eb = (int)((__builtin_bit_cast(unsigned, mx) >> 23) & 0xffu);
    int sb = 259 - eb;
    sb = sb < 1 ? 1 : (sb > 254 ? 254 : sb);
    s = __builtin_bit_cast(float, (unsigned)sb << 23);
  }
  return s;
}
__device__ __forceinline__ float scale_of(const float* p) {      // the same from a recorded maximum (null / zero: 1)
  float s = 1.f;
  const float mx = p ? *p : 0.f;
  if (mx > 0.f) {
    int eb = (int)((__builtin_bit_cast(unsigned, mx) >> 23) & 0xffu);
    int sb = 259 - eb;
    sb = sb < 1 ? 1 : (sb > 254 ? 254 : sb);
    s = __builtin_bit_cast(float, (unsigned)sb << 23);
  }
  return s;
}
// Phi(x), phi(x) from one exp2 and one rcp (A&S 26.2.17; same evaluation as gemm.hip's GEGLU epilogues)
__device__ __forceinline__ void cdf_pdf(float x, float& cdf, float& pdf) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.2316419f, ax, 1.f));
  pdf = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f) * 0.39894228040143267794f;
  float poly = fmaf(1.330274429f, t, -1.821255978f);
  poly = fmaf(poly, t, 1.781477937f);
  poly = fmaf(poly, t, -0.356563782f);
  poly = fmaf(poly, t, 0.319381530f);
  const float q = pdf * (poly * t);
  cdf = x >= 0.f ? 1.f - q : q;
}
// eight scaled floats -> their two fp16 planes (8 halves each): hi = rn(x), lo = rn(x - hi)
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, u32x4& hi, u32x4& lo) {
  const half2v h0 = __builtin_convertvector(f32x2{a[0], a[1]}, half2v), h1 = __builtin_convertvector(f32x2{a[2], a[3]}, half2v);
  const half2v h2 = __builtin_convertvector(f32x2{b[0], b[1]}, half2v), h3 = __builtin_convertvector(f32x2{b[2], b[3]}, half2v);
  const f32x2 r0 = __builtin_convertvector(h0, f32x2), r1 = __builtin_convertvector(h1, f32x2);
  const f32x2 r2 = __builtin_convertvector(h2, f32x2), r3 = __builtin_convertvector(h3, f32x2);
  const half2v l0 = __builtin_convertvector(f32x2{a[0] - r0[0], a[1] - r0[1]}, half2v);
  const half2v l1 = __builtin_convertvector(f32x2{a[2] - r1[0], a[3] - r1[1]}, half2v);
  const half2v l2 = __builtin_convertvector(f32x2{b[0] - r2[0], b[1] - r2[1]}, half2v);
  const half2v l3 = __builtin_convertvector(f32x2{b[2] - r3[0], b[3] - r3[1]}, half2v);
  hi = u32x4{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1), __builtin_bit_cast(unsigned, h2), __builtin_bit_cast(unsigned, h3)};
  lo = u32x4{__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1), __builtin_bit_cast(unsigned, l2), __builtin_bit_cast(unsigned, l3)};
}
// four scaled floats -> two dwords of each plane
__device__ __forceinline__ void split4(const f32x4 a, unsigned& h0, unsigned& h1, unsigned& l0, unsigned& l1) {
  const half2v x0 = __builtin_convertvector(f32x2{a[0], a[1]}, half2v), x1 = __builtin_convertvector(f32x2{a[2], a[3]}, half2v);
  const f32x2 r0 = __builtin_convertvector(x0, f32x2), r1 = __builtin_convertvector(x1, f32x2);
  const half2v y0 = __builtin_convertvector(f32x2{a[0] - r0[0], a[1] - r0[1]}, half2v);
  const half2v y1 = __builtin_convertvector(f32x2{a[2] - r1[0], a[3] - r1[1]}, half2v);
  h0 = __builtin_bit_cast(unsigned, x0); h1 = __builtin_bit_cast(unsigned, x1);
  l0 = __builtin_bit_cast(unsigned, y0); l1 = __builtin_bit_cast(unsigned, y1);
}
__device__ __forceinline__ float amax4(const f32x4 v, float m) {
  m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), m);
  return fmaxf(fmaxf(fabsf(v[2]), fabsf(v[3])), m);
}
// m = max(|a|, |b|, m), pinned where it is written.  A plain fmaxf chain whose result is only read at the end of the kernel
// is sunk by hipcc past the 96-slab loop of the fused feed-forward: the 128 operand values of the tile prologue are SPILLED
// to scratch, reloaded after the loop and reduced there, one scratch reload + vmcnt(0) at a time -- 30 serial memory round
// trips per tile (every LDS-DMA piece in flight included in each wait).
__device__ __forceinline__ void amax_pin(float& m, float a, float b) {
  asm volatile("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(a), "v"(b));
}
__device__ __forceinline__ f32x16 mfma16(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 quad(const f32x16& v, int q) { return f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]}; }

}  // namespace

}  // namespace ramp

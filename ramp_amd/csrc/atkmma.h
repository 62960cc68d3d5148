// 16 x 16 fp16x3 MFMA helpers shared by the sample-owning kernels (atk.hip, atl.hip)
#pragma once
#include "common.h"
#include "tokmma.h"

namespace ramp {
namespace {

constexpr int AT_VG = 1024 + 64;                        // an operand-region group: 4 rows x 256 bytes + padding (bank spread)
constexpr int AT_VW = 12 * AT_VG;                       // per wave and operand (T = 48: 12 groups)

__device__ __forceinline__ f32x4 mm32(const u32x4 a, const u32x4 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
// 2^(target - floor(log2 mx)): mx lands in [2^target, 2^(target + 1)); 1 for mx == 0 / non-finite
__device__ __forceinline__ float pow2_scale(float mx, int target) {
  float s = 1.f;
  if (mx > 0.f && mx < 3.0e38f) {
    const int eb = (int)((__builtin_bit_cast(unsigned, mx) >> 23) & 0xffu);
    int sb = 254 + target - eb;
    sb = sb < 1 ? 1 : (sb > 254 ? 254 : sb);
    s = __builtin_bit_cast(float, (unsigned)sb << 23);
  }
  return s;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// two (scaled) floats -> one dword of each fp16 plane: x s = hi + lo, both rounded to nearest even.  lo comes from v_fma_mix{lo,hi}_f16
// (fp32 fma of x, s and the fp16 hi read in place, rounded to fp16 once): the same bits as (half)(x s - (float)hi) in 4 instructions per
// pair instead of 7 (hipcc does not form the mixed-precision fma itself)
__device__ __forceinline__ void split2m(float x0, float x1, float s, unsigned& hi, unsigned& lo) {
  const half2v h = __builtin_convertvector(f32x2{x0 * s, x1 * s}, half2v);
  hi = __builtin_bit_cast(unsigned, h);
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(l) : "v"(x0), "v"(x1), "v"(s), "v"(hi));
  lo = l;
}
__device__ __forceinline__ void split2m(float x0, float x1, unsigned& hi, unsigned& lo) {
  const half2v h = __builtin_convertvector(f32x2{x0, x1}, half2v);
  hi = __builtin_bit_cast(unsigned, h);
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(l) : "v"(x0), "v"(x1), "v"(hi));
  lo = l;
}
// four floats -> two dwords of each plane
__device__ __forceinline__ void split4m(const f32x4 a, unsigned& h0, unsigned& h1, unsigned& l0, unsigned& l1) {
  split2m(a[0], a[1], h0, l0); split2m(a[2], a[3], h1, l1);
}
// four scaled floats -> two dwords of each plane: x s = hi + lo
__device__ __forceinline__ void split4s(const f32x4 a, float s, u32x2& hi, u32x2& lo) {
  unsigned h0, h1, l0, l1;
  split2m(a[0], a[1], s, h0, l0); split2m(a[2], a[3], s, h1, l1);
  hi = u32x2{h0, h1}; lo = u32x2{l0, l1};
}
__device__ __forceinline__ u32x4 cat2(const u32x2 a, const u32x2 b) { return u32x4{a[0], a[1], b[0], b[1]}; }

}  // namespace
}  // namespace ramp

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
// four scaled floats -> two dwords of each plane: x = hi + lo (both RNE)
__device__ __forceinline__ void split4s(const f32x4 a, float s, u32x2& hi, u32x2& lo) {
  unsigned h0, h1, l0, l1;
  split4(a * s, h0, h1, l0, l1);
  hi = u32x2{h0, h1}; lo = u32x2{l0, l1};
}
__device__ __forceinline__ u32x4 cat2(const u32x2 a, const u32x2 b) { return u32x4{a[0], a[1], b[0], b[1]}; }

}  // namespace
}  // namespace ramp

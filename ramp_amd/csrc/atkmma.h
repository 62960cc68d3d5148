// 16 x 16 fp16x3 MFMA helpers shared by the sample-owning kernels (atk.hip, atl.hip)
#pragma once
#include "core.h"
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
// cross-lane reductions without LDS round trips (ds_bpermute: ~100 cycles of latency each, and these chains are serial).  A 16-lane row by
// DPP (one instruction per step; s_nop 1 = the two wait states between a VALU write and a DPP read of the same register), the four rows by
// v_permlane16_swap / v_permlane32_swap (gfx950): swapping a value with a copy of itself leaves [r0, r0, r2, r2] and [r1, r1, r3, r3] -- the
// two operands of the xor-16 step in every lane; likewise the two halves for xor 32.  Same values, same order as the __shfl_xor chains.
// (inline asm: this toolchain's __builtin_amdgcn_permlane16_swap / 32_swap hands back its first result twice; s_nop 1 on either side = the
// wait states the hazard recogniser puts around the instruction when it emits it itself)
__device__ __forceinline__ void rows_swap(float v, float& a, float& b) {      // a = [r0, r0, r2, r2], b = [r1, r1, r3, r3] of v's 16-lane rows
  a = v; b = v;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void halves_swap(float v, float& a, float& b) {    // a = [lower half, lower half], b = [upper half, upper half]
  a = v; b = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float gmax(float v) {          // max over the lanes c, c + 16, c + 32, c + 48, in all of them
  float a, b;
  rows_swap(v, a, b); v = fmaxf(a, b);
  halves_swap(v, a, b); return fmaxf(a, b);
}
__device__ __forceinline__ float gsum(float v) {          // sum over the lanes c, c + 16, c + 32, c + 48: (own + xor 16) + (the same of xor 32)
  float a, b;
  rows_swap(v, a, b); v = a + b;
  halves_swap(v, a, b); return a + b;
}
__device__ __forceinline__ float wave_max(float v) {      // max over the wave (v >= 0), in every lane
  // (s_nop 4: a VALU write of EXEC -- v_cmpx of a branch around this call -- needs 5 wait states before a DPP instruction, and the hazard
  // recogniser does not look into inline asm; the trailing s_nop 1 covers the permlane swap that reads the result)
  asm("s_nop 4\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
      : "+v"(v));
  return gmax(v);
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
// Between operand planes that split2m / split4s / split4m just wrote and the first MFMA that reads them: the low planes come out of INLINE ASM
// (v_fma_mix*), which the compiler's hazard recogniser does not look into, and a vector write -> MFMA operand read needs two wait states.  With the
// AGPR-form MFMAs of rounds 4-5 the accumulator's zero-initialisation (four v_accvgpr_write) happened to sit in that gap; a VGPR-form MFMA takes 0 as an
// inline constant and follows the split at once -- ato's first query group read stale low planes of v (1e-4 errors on tokens 0-15 of every wave tile).
// One s_nop 1, fenced so that nothing moves across it.
__device__ __forceinline__ void planes_fence() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 1");
  __builtin_amdgcn_sched_barrier(0);
}

}  // namespace
}  // namespace ramp

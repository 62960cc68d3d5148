// Error plumbing, operand-maximum recording and device queries every translation unit of the RAMP sampler HIP library needs (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>

namespace ramp {

// ---- error plumbing -------------------------------------------------------------------------
void set_last_error(const std::string& msg);
const char* last_error_cstr();

#define RAMP_HIP_CHECK(expr)                                                                   \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      ::ramp::set_last_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " at " \
                             + __FILE__ + ":" + std::to_string(__LINE__));                     \
      return -1;                                                                               \
    }                                                                                          \
  } while (0)

#define RAMP_REQUIRE(cond, msg)                                                       \
  do {                                                                                \
    if (!(cond)) {                                                                    \
      ::ramp::set_last_error(std::string("requirement failed: ") + #cond + " — " + (msg) + \
                             " at " + __FILE__ + ":" + std::to_string(__LINE__));     \
      return -2;                                                                      \
    }                                                                                 \
  } while (0)

// [a, a + na) and [b, b + nb) share a byte (null operands never overlap anything)
inline bool ranges_overlap(const void* a, size_t na, const void* b, size_t nb) {
  if (!a || !b || !na || !nb) return false;
  const uintptr_t x = reinterpret_cast<uintptr_t>(a), y = reinterpret_cast<uintptr_t>(b);
  return x < y + nb && y < x + na;
}
// Record a wave's operand maximum.  READ_FIRST: a plain read of the slot, the atomic only where it would raise it -- in a launch of a few
// tens of microseconds whose persistent blocks end together, 2048 same-address atomics serialize in the L2 at ~15 ns each (tkc.hip's
// convolutions back to back: 35 -> 15 us); the read may be stale low (then an unnecessary atomic follows), never high: the slots only grow
// between the zeroing launches of two evaluations.  In the long kernels the dependent read at the tail costs more than the atomics it
// saves (same-box A/B of the whole job: -0.5 % on top of the per-block reduction), so they keep the unconditional atomic.
// What every recording kernel does: ONE atomic per block (record_amax_block) instead of one per wave -- +3.2 % end to end, same box.
#ifdef __HIPCC__
template <bool READ_FIRST = false>
__device__ __forceinline__ void record_amax(float* slot, float amax) {
  if (!slot) return;
  if (READ_FIRST) { if (amax > __builtin_nontemporal_load(slot)) atomicMax(reinterpret_cast<unsigned*>(slot), __builtin_bit_cast(unsigned, amax)); }
  else atomicMax(reinterpret_cast<unsigned*>(slot), __builtin_bit_cast(unsigned, amax));
}
// the 4 waves' maxima of a 256-thread block meet in 16 bytes of LDS (`scratch`: any LDS the block no longer uses), ONE atomic per block
template <bool READ_FIRST = false>
__device__ __forceinline__ void record_amax_block(float* slot, float wave_amax, float* scratch) {
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = wave_amax;
  __syncthreads();
  if (threadIdx.x == 0) record_amax<READ_FIRST>(slot, fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3])));
}
// The same with the range guard of the delayed scale s_in behind it.  Overflow (a scaled element at 60000 or beyond) is judged per
// wave on the rows that wave staged; "the operand shrank" (largest scaled element below 2^-3) on the BLOCK's maximum: the waves of
// the sample-owning kernels hold 4-8 samples each, and per-sample operands (input gradients) legitimately differ by more than the
// 2^8 window from sample to sample -- a wave of small samples must not send the whole job to the bf16x6 kernels.
template <bool READ_FIRST = false>
__device__ __forceinline__ void record_amax_block_guarded(float* slot, float wave_amax, float* scratch, int* range_flag, float s_in, int site) {
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    scratch[threadIdx.x >> 6] = wave_amax;
    if (range_flag && !(wave_amax * s_in < 60000.f)) atomicMax(range_flag, site + 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float bm = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
    record_amax<READ_FIRST>(slot, bm);
    if (range_flag && bm > 0.f && bm * s_in < 0.125f) atomicMax(range_flag, site + 1);
  }
}
#endif
// compute units of the current device (hipDeviceProp_t::multiProcessorCount, cached per device): the token-owning kernels
// launch one 4-wave block per CU
int device_cu_count();
}  // namespace ramp

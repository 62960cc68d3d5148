// Weight / operand packing into MFMA fragment planes (gemm.hip): shared by the tile kernels and the token-owning kernels that stream the same planes.
#pragma once
#include "core.h"

namespace ramp {

int launch_split3(const float* in, unsigned short* out, long n, hipStream_t s);   // fp32 -> 3 bf16 planes
int launch_pack_x6(const float* W, unsigned short* out, long rows, int K, hipStream_t s);   // fp32 [rows][K] -> MFMA-fragment-packed planes
int launch_pack_h3(const float* W, unsigned short* out, long rows, int K, float scale, hipStream_t s);   // same, two fp16 planes

}  // namespace ramp

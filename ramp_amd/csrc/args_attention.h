// Launch arguments of the attention kernels: sample-owning fused blocks (atk.hip, atb.hip, atl.hip) and the stand-alone pair (attention.hip).
#pragma once
#include "core.h"

namespace ramp {

// ---- self-attention fused with the linear layer behind it, sample-owning waves (atk.hip) ---------------------------------------
// forward: Y[m][:] = resid[m][:] + Wo attention(q, k, v)[m][:] + bias + rowbias[rowvar[row0 + m / L]][:]; QKV (M, 768) row-major as the
// QKV linear writes it, 4 heads x 64, softmax over the L tokens of m's sample.  fp16x3 products; the projection's operand o uses
// the delayed scale / maxima / range guard of ONE call site (the out-projection's), the attention-internal operands exact
// per-wave scales.  L must divide 48 or 32 (ato_applicable).
struct AtoArgs {
  int M = 0, L = 0;                    // tokens; tokens per sample
  const float* QKV = nullptr;          // [M][768]
  const unsigned short* W = nullptr;   // the projection's weight stream (ato_pack): 8 slabs x 32 KB
  const float* bias = nullptr;         // [256]
  const float* rowbias = nullptr; const int* rowvar = nullptr; int row0 = 0, rb_stride = 0, n_var = 0;   // <= 4 variants
  const float* resid = nullptr;        // [M][256]
  float* Y = nullptr;                  // [M][256]
  const float* amax_in = nullptr; float* amax_out = nullptr; float wsi = 1.f; int site = 0;
  int* range_flag = nullptr;
  unsigned long long* stamps = nullptr; // diagnostic (ramp_bench_gemm): per wave 8 cycle sums
  int ablate = 0;                      // diagnostic twin (with stamps; wrong results): 2 no k / v DMA, 4 no ring DMA, 6 neither
};
// backward of the attention itself on sample-owning waves (atk.hip): dQKV (M, 768) = d[q | k | v] given dO (M, 256); fp16x3 products with
// exact per-wave operand scales (no call site); same applicability as the forward kernel
struct AtbArgs {
  int M = 0, L = 0;
  const float* QKV = nullptr;          // [M][768]
  const float* dO = nullptr;           // [M][256]
  float* dQKV = nullptr;               // [M][768]
};
int launch_atb(const AtbArgs& a, hipStream_t s);
bool ato_applicable(int M, int L, int* ng);
int launch_ato(const AtoArgs& a, hipStream_t s);
int ato_pack(const float* W /*[256][256] fp32, device*/, float scale, unsigned short* out /*8 * 32 KB*/, hipStream_t s);
int init_atk_attributes();

// ---- attention backward + d(ln1) + LayerNorm-1 backward in one launch of sample-owning waves (atl.hip) ------------------------------
// Y[m][:] = add[m][:] + LNbwd( d(qkv)[m][:] W^T ; Z[m][:], ln_g ),  d(qkv) = attention backward of (QKV, dO): see AtbArgs / TklbArgs.
// d(qkv) is the operand of ONE call site (delayed scale, recorded maximum, range guard); L must divide 48 or 32 (ato_applicable).
struct AblArgs {
  int M = 0, L = 0;
  const float* QKV = nullptr;          // [M][768]
  const float* dO = nullptr;           // [M][256]
  const unsigned short* W = nullptr;   // weight stream (abl_pack): 48 slabs x 16 KB
  const float* Z = nullptr;            // [M][256]: the LayerNorm's input
  const float* add = nullptr;          // [M][256]: the gradient that bypasses the block half
  const float* ln_g = nullptr;         // [256]
  float* Y = nullptr;                  // [M][256]
  const float* amax_in = nullptr; float* amax_out = nullptr; float wsi = 1.f; int site = 0;
  int* range_flag = nullptr;
  unsigned long long* stamps = nullptr; // diagnostic (ramp_bench_gemm): per wave 8 phase sums + 2 totals
  int no_park = 0;                     // diagnostic A/B (ramp_bench_gemm): 1 = the round-4 kernel that fetches k a second time for dQ (same bits)
};
int launch_abl(const AblArgs& a, hipStream_t s);
int abl_pack(const float* W /*[256][768] fp32, device*/, float scale, unsigned short* out /*48 * 16 KB*/, hipStream_t s);
int init_atl_attributes();

// 4-head x 64 softmax self-attention inside each row of L tokens. qkv (R*L, 768) -> o (R*L, 256)
int launch_attn_fwd(const float* qkv, float* o, int R, int L, hipStream_t s);
int launch_attn_bwd(const float* qkv, const float* dout, float* dqkv, int R, int L, hipStream_t s);
int init_attention_attributes();   // raise the dynamic-LDS limit of the stand-alone attention kernels (once)
}  // namespace ramp

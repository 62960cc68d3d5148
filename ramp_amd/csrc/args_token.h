// Launch arguments of the token-owning kernels: fused feed-forward (ffx.hip, ffx16.hip) and the K = 256 linears (tkl.hip, tkl16.hip).
#pragma once
#include "core.h"

namespace ramp {

// ---- fused feed-forward with token-owning waves (ffx.hip) ---------------------------------------
// forward : Y = z2 (M,256) = z1 + W2 (a gelu(g)) + b2, [a | g] = W1 LN3(z1) + b1; writes the VJP stash (private layout)
// backward: Y = dz1 (M,256) = dz + LN3bwd(W1^T [d(hg) s1 | d(hg) s2]; z1), d(hg) = W2^T dz; reads the stash
struct FfxArgs {
  int M = 0;
  const float* X = nullptr;            // forward: z1; backward: dz
  const float* Z1 = nullptr;           // z1 (forward: == X)
  float* Y = nullptr;
  float* stash = nullptr;              // ceil(M / 128) * 128 * 2048 floats, layout private to the two kernels
  const float* ln_g = nullptr; const float* ln_b = nullptr;
  const unsigned short* Wstream = nullptr;   // this direction's weight stream (ffx_build_stream): 96 slabs x 32 KB in consumption order
  const float* b1 = nullptr;           // forward: b1 in the [32 a | 32 g] tiling (2048)
  const float* b2 = nullptr;           // forward: b2 (256)
  const float* amax_in1 = nullptr; float* amax_out1 = nullptr; float wsi1 = 1.f; int site1 = 0;   // first product's operand site
  const float* amax_in2 = nullptr; float* amax_out2 = nullptr; float wsi2 = 1.f; int site2 = 0;   // second product's
  int* range_flag = nullptr;
  unsigned long long* stamps = nullptr; // diagnostic (ablate 64): per wave 4 cycle sums [slab-top wait, barrier, DMA issue, slab body]
  int ablate = 0;                      // diagnostic (ramp_bench_gemm only; wrong results): 1 no LDS-DMA after the first slabs, 2 no stash traffic, 4 no elementwise step, 8 no slab barrier
  // ffx16.hip only: the launch's first token / first stash slot (set by launch_ffx16 for its half-tile launch) and the tiling policy --
  // 0 auto (half tiles for a last round that would idle half of the CUs, and for launches of at most CUs / 2 tiles), 1 full tiles only, 2 half tiles only
  int tok0 = 0, slot0 = 0, half_mode = 0;
};
int launch_ffx(const FfxArgs& f, bool bwd, hipStream_t s);
// W [rows][cols] fp32 -> column-gathered copy (tmp, rows * cols floats) -> fragment-packed fp16 planes (out, 2 * rows * cols halves)
int ffx_pack_second(const float* W, int rows, int cols, int mode, float scale, float* tmp, unsigned short* out, hipStream_t s);
// p1: the first product's fragment-packed fp16 planes (forward: W1 tiled [32 a | 32 g], K = 256; backward: W2^T [1024][256]);
// p2: the second product's, k order permuted by ffx_pack_second (forward: W2 [256][1024]; backward: W1^T [256][2048]);
// out: 96 * 32 KB
int ffx_build_stream(const unsigned short* p1, const unsigned short* p2, unsigned short* out, bool bwd, hipStream_t s);
int init_ffx_attributes();
// the same kernel pair on v_mfma_f32_16x16x32_f16 (ffx16.hip): same arguments, its own weight streams (16 x 32 fragments)
int launch_ffx16(const FfxArgs& f, bool bwd, hipStream_t s);
// W [rows][cols] fp32 -> 16 x 32 fragment planes (tmp: rows * cols floats); perm 0 none, 1 / 2 the k order of the forward / backward second product
int ffx16_pack(const float* W, int rows, int cols, int perm, float scale, float* tmp, unsigned short* out, hipStream_t s);
int ffx16_build_stream(const unsigned short* p1, const unsigned short* p2, unsigned short* out, bool bwd, hipStream_t s);
int init_ffx16_attributes();

// Token-owning linear layer with K = 256 (tkl.hip): Y[m][n] = sum_k pro(X)[m][k] W[n][k] (+ bias[n]) (+ rowbias[rowvar[row0 + m / L]][n])
// (+ resid[m][n]); pro = identity or LayerNorm(256).  fp16x3 products, delayed scale / maxima / range guard of ONE call site.
struct TklArgs {
  int M = 0, N = 0;                    // tokens; output features (multiple of 32, <= 768)
  const float* X = nullptr;            // [M][256]
  float* Y = nullptr; int ldy = 0;
  const unsigned short* W = nullptr;   // fp16 fragment planes of W [N][256] as launch_pack_h3 writes them (= the weight stream: 32 KB per 32 features)
  const float* bias = nullptr;
  const float* resid = nullptr; int ldr = 0;
  const float* rowbias = nullptr; const int* rowvar = nullptr; int row0 = 0, rb_stride = 0, L = 1, n_var = 0;   // N == 256 only
  const float* ln_g = nullptr; const float* ln_b = nullptr;     // LayerNorm over X's 256 columns first (eps 1e-5)
  const float* amax_in = nullptr; float* amax_out = nullptr; float wsi = 1.f; int site = 0;
  int* range_flag = nullptr;
  int ablate = 0;                      // diagnostic (ramp_bench_gemm only)
};
int launch_tkl(const TklArgs& a, hipStream_t s);
// the same linear on v_mfma_f32_16x16x32_f16 (tkl16.hip); W = ffx16_pack(W, N, 256, 0, ...) planes
int launch_tkl16(const TklArgs& a, hipStream_t s);
int init_tkl16_attributes();
int init_tkl_attributes();
// Token-owning d(ln1) with LayerNorm-1 backward in its epilogue (tkl.hip): out = add + LNbwd(X W^T; z, gamma), X = d(qkv) (M, 768),
// W = Wqkv^T as [256][768] (fp16 fragment planes, launch_pack_h3), z = the LayerNorm's input (M, 256), add = the gradient that
// bypasses the block (M, 256).  One call site (the operand X).
struct TklbArgs {
  int M = 0;
  const float* X = nullptr;            // [M][768]
  const float* Z = nullptr;            // [M][256]
  const float* add = nullptr;          // [M][256]
  float* Y = nullptr;                  // [M][256]
  const unsigned short* W = nullptr;   // planes of [256][768]: 96 KB per 32 output features
  const float* ln_g = nullptr;
  const float* amax_in = nullptr; float* amax_out = nullptr; float wsi = 1.f; int site = 0;
  int* range_flag = nullptr;
};
int launch_tklb(const TklbArgs& a, hipStream_t s);
}  // namespace ramp

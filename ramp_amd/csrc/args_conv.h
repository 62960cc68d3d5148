// Launch arguments of the sample-owning k = 5 convolutions (tkc.hip, tkw.hip).
#pragma once
#include "core.h"

namespace ramp {

// ---- Conv1d(k = 5, padding 2) with C_in, C_out in {32, 64} as sample-owning waves (tkc.hip) ---------------------------------------------
// Y[m][n] = sum_tap sum_k X[m + dir (tap - 2)][k] W[tap][n][k] (+ bias[n]) (+ resid[m][n]) (+ resid2[m][n]); rows outside m's sample of L
// tokens read as zero.  dir = +1: the forward convolution; -1: its input gradient (W = the transposed weight, same tap order).  fp16x3
// products, delayed scale / maxima / range guard of ONE call site.  L >= 8 must divide 48 or 32, or be 64 (one sample per wave; not 64 x 64 channels): tkc_applicable.
struct TkcArgs {
  int M = 0, L = 0, N = 0, K = 0, dir = 1;
  const float* X = nullptr; int ldx = 0;
  const unsigned short* W = nullptr;   // tkc_pack: [tap][N / 16][K / 32][plane][lane][8] fp16
  const float* bias = nullptr;
  const float* resid = nullptr; int ldr = 0;
  const float* resid2 = nullptr; int ldr2 = 0;
  float* Y = nullptr; int ldy = 0;
  const float* amax_in = nullptr; float* amax_out = nullptr; float wsi = 1.f; int site = 0;
  int* range_flag = nullptr;
  // round 5, the GroupNorm(8) + Mish around the convolution fused as in tkw.hip (TkwArgs): PRO -- gn_c (M, K) given: the operand is
  // GNbwd(X (.) mish'(gn_gamma x^ + gn_beta) gn_gamma), statistics (M / L, 8, 2) in gn_stats; EPI -- Cst (M, N) given: Cst = conv + bias, its statistics
  // to `stats`, Y = mish(GN(Cst) gamma + beta) + tbias + resid
  const float* gn_c = nullptr; const float* gn_stats = nullptr; const float* gn_gamma = nullptr; const float* gn_beta = nullptr;
  float* Cst = nullptr; float* stats = nullptr; const float* gamma = nullptr; const float* beta = nullptr; const float* tbias = nullptr; float eps = 1e-5f;
};
bool tkc_applicable(int M, int L, int N, int K, int* ng);
int launch_tkc(const TkcArgs& a, hipStream_t s);
int tkc_pack(const float* W /*[5][N][K] fp32, device*/, int N, int K, float scale, unsigned short* out, hipStream_t s);
size_t tkc_packed_halves(int N, int K);
int init_tkc_attributes();

// ---- Conv1d(k = 5, padding 2) with C_out in {128, 256, 512} as sample-owning BLOCKS, GroupNorm(8) + Mish fused around it (tkw.hip) ------------
// Y[m][n] = sum_tap sum_k Xop[m + dir (tap - 2)][k] W[tap][n][k] + bias[n] (+ resid) (+ resid2); rows outside m's sample of L tokens read as zero.
//   operand: Xop = X (channels [0, K1) from X, [K1, K) from X2), or -- gn_c given (PRO 1) -- the GroupNorm + Mish input gradient
//            Xop = GNbwd( X (.) mish'(gn_gamma x^ + gn_beta) gn_gamma ; x^ = (gn_c - mean) rstd ), statistics per (sample, group) in gn_stats;
//   result:  plain (EPI 0, output channels [0, N1) to Y, [N1, N) to Y2), or -- Cst given (EPI 1) -- the convolution output + bias goes to Cst (the
//            VJP stash), its GroupNorm(8) statistics to stats and Y = mish(GN(Cst) gamma + beta) + tbias + resid   (layers.py:280-297, 327-361).
// fp16x3 products, delayed scale / recorded maximum / range guard of ONE call site (the operand Xop).  L >= 3 must divide 96 (tkw_applicable).
struct TkwArgs {
  int M = 0, L = 0, N = 0, K = 0, dir = 1;
  const float* X = nullptr; int ldx = 0;
  const float* X2 = nullptr; int ldx2 = 0; int K1 = 0;       // K1 == K when X2 unused
  const float* gn_c = nullptr; const float* gn_stats = nullptr; const float* gn_gamma = nullptr; const float* gn_beta = nullptr;   // PRO 1; gn_c (M, K)
  const unsigned short* W = nullptr; float wsi = 1.f;       // fp16 fragment planes of W [5][N][K] as launch_pack_h3 writes them
  const float* bias = nullptr;
  const float* resid = nullptr; int ldr = 0;
  const float* resid2 = nullptr; int ldr2 = 0;
  float* Y = nullptr; int ldy = 0;
  float* Y2 = nullptr; int ldy2 = 0; int N1 = 0;            // N1 == N when Y2 unused
  float* Cst = nullptr; float* stats = nullptr;             // EPI 1: (M, N) stash, (M / L, 8, 2) mean / rstd
  const float* gamma = nullptr; const float* beta = nullptr; const float* tbias = nullptr; float eps = 1e-5f;
  const float* amax_in = nullptr; float* amax_out = nullptr; int site = 0;
  int* range_flag = nullptr;
  int ablate = 0;                      // diagnostic (ramp_bench_gemm only; wrong results): 1 no MFMA loop, 2 no operand loads, 4 no epilogue stores
};
bool tkw_applicable(int M, int L, int N, int K, int pro, int epi);
int launch_tkw(const TkwArgs& a, hipStream_t s);
int init_tkw_attributes();
}  // namespace ramp

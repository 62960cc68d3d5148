// Launch arguments of the row-wise, resampling, first / last layer and setup kernels (rowops.hip).
#pragma once
#include "core.h"

namespace ramp {

// ---- row-wise ops (rowops.hip) --------------------------------------------------------------
// GroupNorm over (L, C/8) per (row, group) [+ Mish] [+ per-channel time bias] [+ residual]
struct GnArgs {
  const float* x = nullptr;       // (R, L, C) conv output
  const float* gamma = nullptr; const float* beta = nullptr;
  const float* tbias = nullptr;   // (C) added after the activation, or null
  const float* resid = nullptr;   // (R, L, C) added after the activation, or null
  float* y = nullptr;             // (R, L, C)
  float* stats = nullptr;         // (R, 8, 2) mean, rstd (written)
  int R = 0, L = 0, C = 0; float eps = 1e-5f; int mish = 1;
};
int launch_gn_fwd(const GnArgs& a, hipStream_t s);
// dX of the above: dx = GNbwd( dy * mish'(n) ) (+ add)
struct GnBwdArgs {
  const float* dy = nullptr; const float* x = nullptr; const float* stats = nullptr;
  const float* gamma = nullptr; const float* beta = nullptr;
  const float* add = nullptr;     // (R, L, C) added to the result, or null
  float* dx = nullptr;
  int R = 0, L = 0, C = 0; int mish = 1;
};
int launch_gn_bwd(const GnBwdArgs& a, hipStream_t s);

int launch_ln_fwd(const float* x, const float* gamma, const float* beta, float* y, int n_tok, hipStream_t s);
// dx = add + LNbwd(dy; x, gamma)
int launch_ln_bwd(const float* dy, const float* x, const float* gamma, const float* add, float* dx,
                  int n_tok, hipStream_t s);

// rows of one trajectory leaving / re-entering the shared prefix (rowops.hip): out[r] = in[r / n_rp] (+ rowbias[variant]),
// out[b] = sum_j w[j] in[b n_rp + j]; (rows, L, C) channels-last, w on the host
int launch_expand_rows(const float* in, float* out, int R, int n_rp, int L, int C, const float* rowbias, int rb_stride,
                       const int* rowvar, int row0, hipStream_t s);
int launch_combine_rows(const float* in, float* out, int B, int n_rp, int L, int C, const float* w, hipStream_t s);

// GEGLU on ag (n_tok, 2*F): hg = a * gelu(g); backward writes dag (n_tok, 2*F)
int launch_geglu_fwd(const float* ag, float* hg, int n_tok, int F, hipStream_t s);
int launch_geglu_bwd(const float* dhg, const float* ag, float* dag, int n_tok, int F, hipStream_t s);

// stride-2 resampling convolutions and their dX, one generic gather kernel.
//   mode 0: src = 2*o + j - 1          (Downsample1d fwd, Upsample1d dX)      Lout = Lin/2
//   mode 1: t = o + 1 - j, src = t/2 if t even   (Downsample1d dX, Upsample1d fwd)   Lout = 2*Lin
// W packed as [taps][Cin][Cout] (Cout contiguous). y = bias + sum + add.
struct ResampleArgs {
  const float* x = nullptr; const float* W = nullptr; const float* bias = nullptr; const float* add = nullptr;
  float* y = nullptr; int R = 0, Lin = 0, Lout = 0, Cin = 0, Cout = 0, taps = 3, mode = 0;
};
int launch_resample(const ResampleArgs& a, hipStream_t s);

// first layer: x (B,H,S) -> c1 (R,H,32) [conv k5] and res (R,H,32) [1x1], row r reads x[r / n_rp]
int launch_conv_in_fwd(const float* x, const float* W5 /*[5][S][32]*/, const float* b5, const float* W1 /*[S][32]*/,
                       const float* b1, float* c1, float* res, int R, int n_rp, int H, int S, hipStream_t s);
// eps[r,l,s] = sum_j sum_c dc1[r,l-j+2,c] W5[j][s][c] + sum_c dy[r,l,c] W1[s][c]
int launch_conv_in_bwd(const float* dc1, const float* dy, const float* W5, const float* W1, float* eps,
                       int R, int H, int S, hipStream_t s);
// last layer: f = a Wf^T + bf (R*H, S); da = f Wf  (the seed of the energy gradient: dE/df = f)
int launch_conv_out(const float* a, const float* Wf /*[S][32]*/, const float* bf, float* f, float* da,
                    int n_tok, int S, hipStream_t s);

// ---- setup kernels --------------------------------------------------------------------------
// time-bias table: tb[t][off_i + c] = Wc_i silu(temb(t)) + bc_i for every RTB i, t in [0,T)
struct TimeTableArgs {
  const float* w1; const float* b1; const float* w2; const float* b2;   // time_mlp
  const float* const* cond_w; const float* const* cond_b; const int* couts; const int* offs; int n_rtb;
  float* table; int stride; int T;
  float* temb;          // optional (T, 32): the TimeEncoder output itself (layers.py:233-259), kept for ramp_time_embedding
};
int launch_time_table(const TimeTableArgs& a, hipStream_t s);
// cross-attention bias: out[v][blk][256] = Wo_blk (Wv_blk lat[v]) + bo_blk
int launch_cross_bias(const float* lat, int n_var, int ctx_dim, const float* const* wv, const float* const* wo,
                      const float* const* bo, int n_blk, float* out, hipStream_t s);
}  // namespace ramp

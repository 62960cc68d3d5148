// Launch arguments of the tile GEMM kernels (gemm.hip).
#pragma once
#include "core.h"
#include "pack.h"

namespace ramp {

// ---- GEMM -----------------------------------------------------------------------------------
// C[m, n] = sum_{tap<taps} sum_{k<K} Asrc(m + shift0 + tap*shift_step)[k] * W[tap][n][k]
//           (+ bias[n]) (+ rowbias[rowvar[row0 + m / L]][n]) (+ resid[m, n]) (+ resid2[m, n])
// A rows are tokens of segments ("trajectory rows") of length L; a shifted source row that
// leaves its segment reads as zero (Conv1d zero padding).  The reduction dimension may be
// split over two sources: k < K1 from A, k >= K1 from A2 (channel concat without a copy).
// The output may be split over two destinations: n < N1 to C, n >= N1 to C2.
struct GemmArgs {
  const float* A = nullptr;  int lda = 0;
  const float* A2 = nullptr; int lda2 = 0; int K1 = 0;    // K1 == K when A2 unused
  const float* W = nullptr;                                 // [taps][N][K], K contiguous
  const unsigned short* Wx = nullptr; long wx_plane = 0;    // optional bf16x6 planes [3][taps][N][K] (plane stride in elements)
  // fp16x3 operand scaling (delayed): previous evaluation's max |A| of this call site, where to record this one's,
  // where to flag a scaled operand leaving the fp16 range; 1 / (power-of-two scale the packed weights carry)
  const float* a_absmax_in = nullptr; float* a_absmax_out = nullptr; int* range_flag = nullptr; float w_scale_inv = 1.f;
  int site_id = 0;
  int geglu_group = 64;                                     // EPI_GEGLU_FWD weight tiling: [group a-rows | group g-rows]
  const float* Amul = nullptr; int lda_mul = 0; int a_period = 0;   // optional: A_eff[m][k] = A[m][k % a_period] * Amul[m][k]
  int ablate = 0;                                           // diagnostic kernel variant (ramp_bench_gemm only)
  int three_ok = 1;                                         // launch plan: a third resident block where it measured faster
  int tile_pref = 0;                                        // tuning override (micro-benchmarks): 0 auto, 1 force the 128 x 128 tile, 3 force 3 blocks / CU
  int wx_packed = 0;                                        // 1: Wx is fragment-packed [taps*N/32][K/16][3][64][8] bf16 (launch_pack_x6); 2: [..][2][64][8] fp16 (launch_pack_h3)
  const float* bias = nullptr;                              // [N]
  const float* rowbias = nullptr; const int* rowvar = nullptr; int row0 = 0; int rb_stride = 0;  // [n_var][rb_stride]
  const float* resid = nullptr;  int ldr = 0;
  const float* resid2 = nullptr; int ldr2 = 0;
  float* C = nullptr;  int ldc = 0;
  float* C2 = nullptr; int ldc2 = 0; int N1 = 0;            // N1 == N when C2 unused
  int M = 0, N = 0, K = 0;
  int taps = 1, shift0 = 0, shift_step = 0, L = 1;
  // strided rows: source row of output token (seg, l) is seg*(L*a_stride) + l*a_stride + shift (stride-2
  // convs); output/residual row is m*c_rstride + c_roff (the even / odd phases of a transposed conv)
  int a_stride = 1, c_rstride = 1, c_roff = 0;
  // fused epilogues
  int epi = 0;                       // EPI_*
  const float* aux_in = nullptr;     // EPI_GEGLU_BWD: ag (M, 2N)
  float* aux_out = nullptr;          // EPI_GEGLU_FWD: hg (M, N/2)
  int ld_aux = 0;
};
// EPI_LINEAR:    C = acc + bias + rowbias + resid + resid2
// EPI_GEGLU_FWD: weights packed so that each 128-column tile is [64 a-columns | the 64 matching g-columns];
//                with [a | g] = acc + bias: aux_out (M, N/2) = a * gelu(g), and C (M, N) receives the VJP stash
//                [gelu(g) | a * gelu'(g)]
// EPI_GEGLU_BWD: acc = d(hg) (M, N); C (M, 2N) = d[a | g] = [acc * s1 | acc * s2] with the stash [s1 | s2] in aux_in
enum { EPI_LINEAR = 0, EPI_GEGLU_FWD = 1, EPI_GEGLU_BWD = 2 };
int launch_gemm(const GemmArgs& a, hipStream_t s);
int launch_ff_fwd(const GemmArgs& ff1, const GemmArgs& ff2, hipStream_t s);   // fused FF1 -> GEGLU -> FF2 (gemm.hip)
int init_gemm_attributes();        // raise the dynamic-LDS limit of every GEMM instantiation (once)
}  // namespace ramp

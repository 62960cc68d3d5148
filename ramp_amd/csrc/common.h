// Shared declarations for the RAMP sampler HIP library (gfx950 / MI355X only).
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
int launch_split3(const float* in, unsigned short* out, long n, hipStream_t s);   // fp32 -> 3 bf16 planes
int launch_pack_x6(const float* W, unsigned short* out, long rows, int K, hipStream_t s);   // fp32 [rows][K] -> MFMA-fragment-packed planes
int launch_pack_h3(const float* W, unsigned short* out, long rows, int K, float scale, hipStream_t s);   // same, two fp16 planes
int launch_ff_fwd(const GemmArgs& ff1, const GemmArgs& ff2, hipStream_t s);   // fused FF1 -> GEGLU -> FF2 (gemm.hip)
int init_gemm_attributes();        // raise the dynamic-LDS limit of every GEMM instantiation (once)
int init_attention_attributes();   // same for the attention kernels

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

// ---- Conv1d(k = 5, padding 2) with C_in, C_out in {32, 64} as sample-owning waves (tkc.hip) ---------------------------------------------
// Y[m][n] = sum_tap sum_k X[m + dir (tap - 2)][k] W[tap][n][k] (+ bias[n]) (+ resid[m][n]) (+ resid2[m][n]); rows outside m's sample of L
// tokens read as zero.  dir = +1: the forward convolution; -1: its input gradient (W = the transposed weight, same tap order).  fp16x3
// products, delayed scale / maxima / range guard of ONE call site.  L >= 8 must divide 48 or 32 (tkc_applicable).
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

// 4-head x 64 softmax self-attention inside each row of L tokens. qkv (R*L, 768) -> o (R*L, 256)
int launch_attn_fwd(const float* qkv, float* o, int R, int L, hipStream_t s);
int launch_attn_bwd(const float* qkv, const float* dout, float* dqkv, int R, int L, hipStream_t s);

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

// ---- sampler (sampler.hip) ------------------------------------------------------------------
struct CfgMeanArgs {
  const float* x = nullptr;     // (B,H,S)
  const float* eps = nullptr;   // (B*n_rp,H,S) interleaved [v0,v1,(v2)] per trajectory
  float* x0 = nullptr; float* mean = nullptr;   // either may be null
  float* ecomb = nullptr;       // optional
  int B = 0, HS = 0, n_rp = 2;
  float w0 = 0, w1 = 0, w0p1 = 1; // n_rp=2: e=w0p1*v0 - w0*v1 (w0p1 = float(1+w)) ; n_rp=3: e=v2+w0*(v0-v2)+w1*(v1-v2)
  float sqrt_recip = 0, sqrt_recipm1 = 0, coef1 = 0, coef2 = 0; int clip = 1;
  int predict_x0 = 0;           // predict_epsilon=False (the reference constructor's default): the combined network output IS x0
};
int launch_cfg_mean(const CfgMeanArgs& a, hipStream_t s);

struct HardConds { const int* idx = nullptr; const float* val = nullptr; int n = 0; };  // val (n,B,S)

// x = mean + (std * z) * noise_scale ; z = 0 when !use_noise (t == 0) ; then hard conditioning
int launch_ddpm_finish(const float* mean, const float* noise, float stdv, float noise_scale, int use_noise,
                       HardConds hc, float* x, float* chain_out, int B, int H, int S, hipStream_t s);
int launch_ddim_finish(const float* x_in, const float* x0, float sqrt_a_t, float sqrt_1m_a_t, float sqrt_a_prev,
                       float dir_coef, HardConds hc, float* x, float* chain_out, int B, int H, int S, hipStream_t s);
int launch_hard_cond(float* x, HardConds hc, int B, int H, int S, hipStream_t s);
// out[0..n) ~ N(0, 1): Philox4x32-10 + Box-Muller, rec = device {seed, offset in groups of four elements} (sampler.hip)
int launch_philox_normal(float* out, long n, const unsigned long long* rec, hipStream_t s);
// the same stream addressed by GLOBAL sample index: out is this shard's (n_blocks, B, HS) noise block of a job whose whole
// noise block is (n_blocks, B_total, HS); local sample b is global sample sample0 + b, i.e. out[(j B + b) HS + e] = element
// (j B_total + sample0 + b) HS + e of the stream (HS % 4 == 0).  B_total == B, sample0 == 0 is launch_philox_normal.
int launch_philox_normal_sharded(float* out, int n_blocks, int B, int HS, long sample0, long B_total, const unsigned long long* rec, hipStream_t s);

struct ApfArgs {
  float* traj = nullptr;        // (B,H,S) modified in place (xy channels only)
  const float* cloud = nullptr; // (P,2)
  const float* window = nullptr;// (2*win+1) Gaussian weights
  int B = 0, H = 0, S = 0, P = 0, win = 0;
  double thr = 0, strength = 0;
};
int launch_apf(const ApfArgs& a, hipStream_t s);
struct ApfDynArgs {
  float* traj = nullptr;          // (B,H,S) in place (xy only)
  const double* points = nullptr; // (P,2) float64
  const float* goal = nullptr;    // (S) goal state for the pursuer pass, or null
  const int* enable = nullptr;    // (B) per-trajectory switch, or null = all
  int B = 0, H = 0, S = 0, P = 0;
  int window = -1;                // >= 0: static pass around the closest waypoint; < 0: waypoints [0, affected)
  int affected = 0;
  double thr_query = 0, thr_force = 0, strength = 0;
};
int launch_apf_dynamic(const ApfDynArgs& a, hipStream_t s);
// receding-horizon replanning (sampler.hip): what changes from replan to replan, resident on the device
struct ReplanState { int n_hist; int stepp; int pad0; int pad1; float pursuer[2]; float pad2[2]; };
int launch_replan_init(float* x, const float* x_clean, const float* noise, float sa, float s1a, const float* hist,
                       const ReplanState* st, int B, int H, int S, hipStream_t s);
int launch_replan_pin(float* x, HardConds hc, const float* hist, const float* x_clean, const ReplanState* st, int B, int H,
                      int S, hipStream_t s);
int launch_replan_sm(float* x, const ReplanState* st, int window, float dt, float max_vel, int B, int H, int S, hipStream_t s);
int launch_replan_near(const float* x, const ReplanState* st, float thr, int* en, int B, int H, int S, hipStream_t s);
int launch_replan_goal(float* x0, const float* x, int B, int H, int S, hipStream_t s);
int launch_replan_select(const float* traj, const int* mask, const float* plen, const float* smooth, float w_s, float w_l,
                         float* best, int* result, int B, int H, int S, hipStream_t s);
// mask[b] = any_{h,p} ||xy - p|| < thr ; plen[b], smooth[b]
int launch_traj_costs(const float* traj, const float* cloud, int B, int H, int S, int P, float thr,
                      int* mask, float* plen, float* smooth, hipStream_t s);
int launch_traj_metrics(const float* traj, int B, int H, int S, const float* centers, const float* sizes, int n_boxes,
                        float* intensity, float* path_len, float* smooth, hipStream_t s);
// scratch: 2 * H * ceil(B / 256) doubles; out: 1 double
int launch_waypoint_variance(const float* traj, int B, int H, int S, double* scratch, double* out, hipStream_t s);

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

// ---- scene encoders (scene.hip) ----------------------------------------------------------------
int scene_enc2d_prep(const float* cloud, int No, int Np, float* centers, float* maxd, hipStream_t s);
int scene_enc2d_feat(const float* cloud, const float* centers, const float* maxd, const float* div, const float* w0,
                     const float* b0, const float* g0, const float* be0, float* feat, int Np, int T, hipStream_t s);
int scene_ln64(const float* x, const float* g, const float* b, float* y, int T, int act, hipStream_t s);
int scene_affine_act(const float* x, const float* scale, const float* shift, const float* add, float* y, long n, int C,
                     int act, hipStream_t s);     // act: 0 none, 1 GELU, 2 SELU
int scene_bn_fold(const float* w, const float* b, const float* rm, const float* rv, float* scale, float* shift, int C, hipStream_t s);
int scene_linear_small(const float* x, const float* W, const float* b, float* y, long T, int N, int K, hipStream_t s);
int scene_colreduce(const float* x, float* out, int n_seg, int seglen, int C, int mode /*0 mean, 1 max*/, hipStream_t s);
int scene_attention(const float* qkv, float* o, int T, int heads, int dh, float scale, hipStream_t s);

}  // namespace ramp

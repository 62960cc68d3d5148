// fp32 MFMA GEMM with conv taps for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains).
//
// Serves every dense contraction on the sampler hot path: the transformer linears
// (layers_attention_mini.py:83-127, 38-45), Conv1d k=5 / k=1 and the stride-2 resampling convs as
// shifted-tap GEMMs in the channels-last layout (layers.py:262-297, 327-361) and all of their dX
// products (UnetInference.py:27 — autograd.grad w.r.t. the input only, weights are frozen).
//
// Structure
//  * Tile BM x BN x 32, 4 waves, each wave owns (MI*32) x (NI*32) as MI*NI 32x32 accumulators.
//    Both operands are K-contiguous ([M][K] activations, [N][K] weights = nn.Linear layout),
//    staged global -> registers -> LDS with one 16-byte pad per row, so every lane feeds four
//    MFMAs from a single conflict-free ds_read_b128 per operand (the k order inside a 32-chunk is
//    permuted identically for A and B).  Double-buffered LDS, one barrier per K-tile.
//  * PERSISTENT blocks: each XCD owns a contiguous run of logical tiles (N fastest) and its resident blocks
//    walk it interleaved, so the N-tiles of one M-tile are in flight together and hit the same L2.  The global loads of the
//    next tile's first K-slab are issued before the current tile's epilogue and parked in
//    registers, so the short-K shapes of this network (K = 256) do not pay a cold prologue per tile.
//  * Epilogue through LDS: accumulators are transposed in LDS and leave as coalesced float4 rows;
//    bias, per-row-variant bias, up to two residuals, and the GEGLU forward / backward elementwise
//    math are applied there (no separate elementwise passes over the 2048-wide hidden).
#include "args_gemm.h"

#include <algorithm>

namespace ramp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;   // floats per LDS row (16-byte pad: conflict-free ds_read_b128)

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
  const float pdf = expf(-0.5f * x * x) * 0.39894228040143267794f;
  return cdf + x * pdf;
}

// Phi(x) and phi(x) of the standard normal from ONE hardware exp2 and one rcp: Q(|x|) = phi(x) * t * P(t),
// t = 1 / (1 + p |x|) (Abramowitz & Stegun 26.2.17, |error| < 7.5e-8; 3e-7 absolute once evaluated in fp32, i.e. the
// rounding level of the erf form 0.5 (1 + erf(x / sqrt 2)) itself, which cancels just as badly in the left tail).
// ~18 VALU instructions per element instead of ~55 for erff + expf: the GEGLU-forward epilogue handles 32 (a, g)
// pairs per thread per tile and was bound by the vector issue port, not by the matrix pipe or by HBM.
__device__ __forceinline__ void normal_cdf_pdf(float x, float& cdf, float& pdf) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.2316419f, ax, 1.f));
  pdf = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f) * 0.39894228040143267794f;
  float poly = fmaf(1.330274429f, t, -1.821255978f);
  poly = fmaf(poly, t, 1.781477937f);
  poly = fmaf(poly, t, -0.356563782f);
  poly = fmaf(poly, t, 0.319381530f);
  const float q = pdf * (poly * t);
  cdf = x >= 0.f ? 1.f - q : q;
}

// Accumulators -> LDS -> coalesced float4 rows, with the fused epilogue math.  The caller guarantees that no
// wave still reads operand tiles from `smem` (a barrier precedes the call); ends with the C tile fully consumed
// by this thread's own reads only (the caller issues the next barrier before reusing `smem`).
// HALVES = WM: the C image holds one wave-row (MI * 32 tile rows) at a time (the pipelined x6 kernel lends it
// only one operand buffer); the caller then needs no barrier before the call but one after it.
template <int WM, int WN, int MI, int NI, int EPI, bool GEN, int HALVES = 1, bool OSC = false>
__device__ __forceinline__ void epilogue(const GemmArgs& a, f32x16 (&acc)[MI][NI], float* smem, int tile, int tiles_n,
                                         int tid, int wm, int wn, int r, int h, const float oscale = 1.f) {
  static_assert(HALVES == 1 || HALVES == WM, "C image: whole tile or one wave-row at a time");
  constexpr int BMT = WM * MI * 32, BM = BMT / HALVES, BN = WN * NI * 32, NT = WM * WN * 64;
  constexpr int CLD = BN + 4;
  // ---------------- epilogue: accumulators -> LDS -> coalesced float4 rows ----------------
  const int tile_m = tile / tiles_n;
  const int n0 = (tile - tile_m * tiles_n) * BN;
  float* Cs = smem;
#pragma unroll 1
  for (int g = 0; g < HALVES; ++g) {
  const int m0 = tile_m * BMT + g * BM;
  // The global operands of the fused math (residual rows / GEGLU stash) do not depend on the C image: the first
  // chunk is requested BEFORE the accumulators go through LDS, so its HBM latency hides behind the transpose.
  constexpr int C4N = BN / 4, RP = NT / C4N, PASSES = BM / RP;
  constexpr int CHMAX = EPI == EPI_GEGLU_BWD ? 4 : 8;
  constexpr int CH = PASSES < CHMAX ? PASSES : CHMAX;
  static_assert(PASSES % CH == 0, "epilogue chunking");
  const int cc = tid % C4N, rr = tid / C4N;
  const int n = n0 + cc * 4;
  const bool nok = n < a.N;
  const int nc = nok ? n : 0;
  f32x4 pre0[CH], pre1[EPI == EPI_GEGLU_BWD ? CH : 1];
  auto prefetch = [&](int c0) {
    if (EPI == EPI_LINEAR) {
      if (a.resid) {
#pragma unroll
        for (int q = 0; q < CH; ++q) {
          const int m = m0 + (c0 + q) * RP + rr;
          const int mc = m < a.M ? m : 0;
          const long orow = GEN ? (long)mc * a.c_rstride + a.c_roff : (long)mc;
          pre0[q] = *reinterpret_cast<const f32x4*>(a.resid + orow * a.ldr + nc);
        }
      }
    } else if (EPI == EPI_GEGLU_BWD) {
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        const int m = m0 + (c0 + q) * RP + rr;
        const int mc = m < a.M ? m : 0;
        pre0[q] = *reinterpret_cast<const f32x4*>(a.aux_in + (long)mc * a.ld_aux + nc);
        pre1[q] = *reinterpret_cast<const f32x4*>(a.aux_in + (long)mc * a.ld_aux + a.N + nc);
      }
    }
  };
  prefetch(0);
  if (HALVES == 1 || wm == g) {
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = (HALVES == 1 ? wm * MI * 32 : 0) + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        Cs[row * CLD + wn * NI * 32 + ni * 32 + r] = acc[mi][ni][reg];
      }
  }
  __syncthreads();
  if (EPI == EPI_LINEAR || EPI == EPI_GEGLU_BWD) {
    f32x4 b4 = {0, 0, 0, 0};
    if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + nc);
#pragma unroll 1
    for (int c0 = 0; c0 < PASSES; c0 += CH) {
      if (c0) prefetch(c0);
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        const int row = (c0 + q) * RP + rr;
        const int m = m0 + row;
        const bool ok = nok && m < a.M;
        const int mc = m < a.M ? m : 0;
        const long orow = GEN ? (long)mc * a.c_rstride + a.c_roff : (long)mc;
        f32x4 v = *reinterpret_cast<const f32x4*>(Cs + row * CLD + cc * 4);
        if (OSC) v *= oscale;                          // undo the operand scales (powers of two: exact)
        v += b4;
        if (EPI == EPI_LINEAR) {
          if (a.rowbias) v += *reinterpret_cast<const f32x4*>(a.rowbias + (long)a.rowvar[a.row0 + mc / a.L] * a.rb_stride + nc);
          if (a.resid) v += pre0[q];
          if (GEN && a.resid2) v += *reinterpret_cast<const f32x4*>(a.resid2 + orow * a.ldr2 + nc);
          if (ok) {
            if (!GEN || n < a.N1) *reinterpret_cast<f32x4*>(a.C + orow * a.ldc + n) = v;
            else *reinterpret_cast<f32x4*>(a.C2 + orow * a.ldc2 + (n - a.N1)) = v;
          }
        } else {
          // v = d(hg)[m][n]; the forward stashed s = [gelu(g) | a * gelu'(g)] (n_tok, 2N): d(ag) = [v * s1 | v * s2]
          const f32x4 da = v * pre0[q], dg = v * pre1[q];
          if (ok) {
            *reinterpret_cast<f32x4*>(a.C + (long)m * a.ldc + n) = da;
            *reinterpret_cast<f32x4*>(a.C + (long)m * a.ldc + a.N + n) = dg;
          }
        }
      }
    }
  } else {   // EPI_GEGLU_FWD: tile columns [0, BN/2) hold a_j, [BN/2, BN) the matching g_j (weights packed so)
    constexpr int C4N = BN / 8, RP = NT / C4N, PASSES = BM / RP;
    const int cc = tid % C4N, rr = tid / C4N;
    const int half = a.N >> 1;
    const int j = (n0 >> 1) + cc * 4;                // index inside the a- (and g-) half
    f32x4 ba = {0, 0, 0, 0}, bg = {0, 0, 0, 0};
    if (a.bias) {
      ba = *reinterpret_cast<const f32x4*>(a.bias + n0 + cc * 4);
      bg = *reinterpret_cast<const f32x4*>(a.bias + n0 + BN / 2 + cc * 4);
    }
#pragma unroll 4
    for (int p = 0; p < PASSES; ++p) {
      const int row = p * RP + rr;
      const int m = m0 + row;
      f32x4 av = *reinterpret_cast<const f32x4*>(Cs + row * CLD + cc * 4);
      f32x4 gv = *reinterpret_cast<const f32x4*>(Cs + row * CLD + BN / 2 + cc * 4);
      if (OSC) { av *= oscale; gv *= oscale; }
      av += ba; gv += bg;
      f32x4 hv, s1, s2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float cdf, pdf;
        normal_cdf_pdf(gv[e], cdf, pdf);
        s1[e] = gv[e] * cdf;                           // gelu(g)
        s2[e] = av[e] * (cdf + gv[e] * pdf);           // a * gelu'(g)
        hv[e] = av[e] * s1[e];                         // hg = a * gelu(g)
      }
      if (m < a.M) {
        *reinterpret_cast<f32x4*>(a.C + (long)m * a.ldc + j) = s1;            // stash for the VJP: no transcendental
        *reinterpret_cast<f32x4*>(a.C + (long)m * a.ldc + half + j) = s2;     // is needed in the backward epilogue
        *reinterpret_cast<f32x4*>(a.aux_out + (long)m * a.ld_aux + j) = hv;
      }
    }
  }
  if (HALVES > 1) __syncthreads();      // the image is reused by the next half / by the caller's operand stores
  }
}

// GEN = false: plain linear (1 tap, unit strides, single source / destination, no second residual), the hot
//               transformer shapes; operand rows outside M / N are clamped, never masked (they only feed
//               outputs that are never stored) and K % 32 == 0, so the loader is 8 unconditional 16-byte loads.
// GEN = true:  conv taps with zero padding at segment ends, stride-2 sources, split-K source (channel concat),
//               split-N / strided destination, second residual.
template <int WM, int WN, int MI, int NI, int EPI, bool GEN>
__global__ __launch_bounds__(WM * WN * 64)
void gemm_kernel(GemmArgs a, int tiles_n, int n_tiles) {
  constexpr int BM = WM * MI * 32, BN = WN * NI * 32, NT = WM * WN * 64;
  constexpr int AI = BM * 8 / NT, BI = (BN * 8 + NT - 1) / NT;   // float4 loads per thread per tile
  constexpr int CLD = BN + 4;                                      // C staging row stride (floats)
  static_assert(BM * 8 % NT == 0, "tile/thread mismatch");
  static_assert(BM * CLD <= 2 * (BM + BN) * LDS_LD, "C staging must fit in the operand stages");
  extern __shared__ __attribute__((aligned(16))) float smem[];

  // ---- persistent tile range of this block (XCD-aware) ----
  const int bid = blockIdx.x, nb = gridDim.x;          // nb is a multiple of 8
  const int xcd = bid & 7, slot = bid >> 3, bpx = nb >> 3;
  const long xlo = (long)xcd * n_tiles / 8, xhi = (long)(xcd + 1) * n_tiles / 8;
  // Interleaved assignment inside the XCD's run: at any moment the resident blocks of one XCD work on
  // bpx CONSECUTIVE logical tiles (= a few M-tiles x all their N-tiles), so the A tiles they share and the
  // weight panel stay in that XCD's 4 MiB L2 (a contiguous run per block re-fetched A once per N-tile).
  const int t_begin = (int)xlo + slot, t_end = (int)xhi, t_step = bpx;
  if (t_begin >= t_end) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int c4 = tid & 7, r0 = tid >> 3;               // staging: float4 column, first row
  constexpr int RSTEP = NT / 8;

  float* As0 = smem;
  float* Bs0 = smem + BM * LDS_LD;
  constexpr int STAGE = (BM + BN) * LDS_LD;

  const int nk = a.K / BK;                             // K % 32 == 0 (checked on the host)
  const int total = GEN ? a.taps * nk : nk;

  // ---- loader state: (tile, k-iteration) one step ahead of the MFMA loop ----
  int ld_tile = t_begin, ld_it = 0;
  const float* ap[AI];                                 // row base pointers (+ float4 column) of this thread
  const float* bp[BI];
  int a_l[GEN ? AI : 1]; long a_m[GEN ? AI : 1];       // GEN: in-segment position / source row (no shift)
  const int Lin = GEN ? a.L * a.a_stride : 0;
  auto setup_rows = [&](int tile) {
    const int tile_m = tile / tiles_n;
    const int n0 = (tile - tile_m * tiles_n) * BN;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      int m = tile_m * BM + r0 + RSTEP * i;
      if (!GEN) {
        m = m < a.M ? m : a.M - 1;
        ap[i] = a.A + (long)m * a.lda + c4 * 4;
      } else {
        const bool mv = m < a.M;
        m = mv ? m : 0;
        const int seg = m / a.L, l = m - seg * a.L;
        a_l[i] = mv ? l * a.a_stride : -(1 << 28);    // invalid rows fail every range test
        a_m[i] = (long)seg * Lin + l * a.a_stride;
      }
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      int n = n0 + r0 + RSTEP * i;
      n = n < a.N ? n : a.N - 1;
      bp[i] = a.W + (long)n * a.K + c4 * 4;
    }
  };
  f32x4 ra[AI], rb[BI];
  auto load_tile = [&]() {                             // straight-line: every address is valid
    if (!GEN) {
      const int k0 = ld_it * BK;
#pragma unroll
      for (int i = 0; i < AI; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap[i] + k0);
#pragma unroll
      for (int i = 0; i < BI; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bp[i] + k0);
    } else {
      const int tap = ld_it / nk, k0 = (ld_it - tap * nk) * BK;
      const int sh = a.shift0 + tap * a.shift_step;
      const float* Ab; int ld, kc;
      if (k0 < a.K1) { Ab = a.A; ld = a.lda; kc = k0; } else { Ab = a.A2; ld = a.lda2; kc = k0 - a.K1; }
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const int l = a_l[i] + sh;
        const bool ok = l >= 0 && l < Lin;
        const long row = ok ? a_m[i] + sh : a_m[i];
        const f32x4 v = *reinterpret_cast<const f32x4*>(Ab + row * ld + kc + c4 * 4);
        ra[i] = ok ? v : f32x4{0, 0, 0, 0};
      }
      const long woff = (long)tap * a.N * a.K + k0;
#pragma unroll
      for (int i = 0; i < BI; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bp[i] + woff);
    }
  };
  auto store_tile = [&](int st) {
    float* As = As0 + st * STAGE;
    float* Bs = Bs0 + st * STAGE;
#pragma unroll
    for (int i = 0; i < AI; ++i)
      *reinterpret_cast<f32x4*>(As + (r0 + RSTEP * i) * LDS_LD + c4 * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < BI; ++i)
      if (BN * 8 % NT == 0 || r0 + RSTEP * i < BN) *reinterpret_cast<f32x4*>(Bs + (r0 + RSTEP * i) * LDS_LD + c4 * 4) = rb[i];
  };
  auto advance_loader = [&]() {          // after the last tile the loader idles on it (loads are harmless)
    if (++ld_it == total) {
      ld_it = 0;
      if (ld_tile + t_step < t_end) { ld_tile += t_step; setup_rows(ld_tile); }
    }
  };

  const int r = lane & 31, h = lane >> 5;
  const int arow = (wm * MI * 32 + r) * LDS_LD + h * 4;
  const int brow = (wn * NI * 32 + r) * LDS_LD + h * 4;

  setup_rows(ld_tile);
  load_tile();
  store_tile(0);
  advance_loader();
  __syncthreads();
  int st = 0;
  for (int tile = t_begin; tile < t_end; tile += t_step) {
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    for (int it = 0; it < total; ++it) {
      load_tile();                       // next slab (of this or the next tile) into registers
      __builtin_amdgcn_sched_barrier(0); // keep the global loads ahead of the MFMA block
      const float* As = As0 + st * STAGE;
      const float* Bs = Bs0 + st * STAGE;
      f32x4 av[4][MI], bv[4][NI];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          av[q][mi] = *reinterpret_cast<const f32x4*>(As + arow + mi * 32 * LDS_LD + q * 8);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          bv[q][ni] = *reinterpret_cast<const f32x4*>(Bs + brow + ni * 32 * LDS_LD + q * 8);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][mi][e], bv[q][ni][e], acc[mi][ni], 0, 0, 0);
      if (it + 1 < total) { store_tile(st ^ 1); advance_loader(); }   // else: parked in registers over the epilogue
      __syncthreads();
      st ^= 1;
    }

    epilogue<WM, WN, MI, NI, EPI, GEN>(a, acc, smem, tile, tiles_n, tid, wm, wn, r, h);
    __syncthreads();
    if (tile + t_step < t_end) { store_tile(st); advance_loader(); __syncthreads(); }
  }
}

// =================================================================================================
// bf16x6 variant: fp32-accurate products on the bf16 matrix cores.
//
// Every fp32 operand is split into three bf16 planes x = x1 + x2 + x3 (8 + 8 + 8 significand bits, residual
// <= 2^-25 |x|); a*b is accumulated in fp32 as a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1 (each bf16*bf16 product
// is exact in fp32; the dropped terms a2b3 + a3b2 + a3b3 are <= 2^-25 |ab|, i.e. below fp32 rounding).  Six
// v_mfma_f32_32x32x16_bf16 (16x the fp32-MFMA rate) replace eight v_mfma_f32_32x32x2_f32: 2.7x less matrix-core
// time at fp32-level accuracy.  Weights are split once at load (planes [3][taps][N][K] bf16); activations are
// split while they are staged into LDS (once per element, by the staging thread).
// Single LDS stage (3 A planes + 3 B planes; rows of 64 bytes whose four 16-byte slots are XOR-swizzled with
// (row >> 2) & 3, which makes the b64 / b128 staging writes and the ds_read_b128 fragment reads conflict-free),
// two blocks per CU; the next slab always waits in registers (global loads in flight during the MFMAs and the epilogue).
// =================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int XLD = 32;          // bf16 elements per LDS row (64 bytes, unpadded; 16-byte slots XOR-swizzled by (row >> 2) & 3)

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
// x (4 floats) -> three planes of 4 bf16 (2 dwords each)
__device__ __forceinline__ void split3(const f32x4 x, u32x2& p1, u32x2& p2, u32x2& p3) {
  f32x4 r1, r2;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const unsigned q = pk_bf16(x[2 * e], x[2 * e + 1]);
    p1[e] = q;
    r1[2 * e] = x[2 * e] - __builtin_bit_cast(float, q << 16);
    r1[2 * e + 1] = x[2 * e + 1] - __builtin_bit_cast(float, q & 0xffff0000u);
  }
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const unsigned q = pk_bf16(r1[2 * e], r1[2 * e + 1]);
    p2[e] = q;
    r2[2 * e] = r1[2 * e] - __builtin_bit_cast(float, q << 16);
    r2[2 * e + 1] = r1[2 * e + 1] - __builtin_bit_cast(float, q & 0xffff0000u);
  }
  p3[0] = pk_bf16(r2[0], r2[1]);
  p3[1] = pk_bf16(r2[2], r2[3]);
}

__global__ void split3_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, long n4, long plane) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(in)[i];
    u32x2 p1, p2, p3;
    split3(v, p1, p2, p3);
    reinterpret_cast<u32x2*>(out)[i] = p1;
    reinterpret_cast<u32x2*>(out + plane)[i] = p2;
    reinterpret_cast<u32x2*>(out + 2 * plane)[i] = p3;
  }
}
int launch_split3(const float* in, unsigned short* out, long n, hipStream_t s) {
  RAMP_REQUIRE(n > 0 && n % 4 == 0, "split3 needs a multiple of 4 elements");
  const long n4 = n / 4;
  hipLaunchKernelGGL(split3_kernel, dim3((int)std::min<long>((n4 + 255) / 256, 8192)), dim3(256), 0, s, in, out, n4, n);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

template <int EPI, bool GEN>
__global__ __launch_bounds__(256)
void gemm_x6_kernel(GemmArgs a, int tiles_n, int n_tiles) {
  constexpr int WM = 2, WN = 2, MI = 2, NI = 2;
  constexpr int BM = 128, BN = 128, NT = 256;
  constexpr int AI = 4;                               // fp32 float4 loads per thread (A)
  constexpr int BI = 2;                               // 16-byte bf16 loads per thread per plane (B)
  constexpr int PLANE = BM * XLD;                     // bf16 elements per LDS plane
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short* Ax = reinterpret_cast<unsigned short*>(smem);           // [3][BM][XLD]
  unsigned short* Bx = Ax + 3 * PLANE;                                     // [3][BN][XLD]

  const int bid = blockIdx.x, nb = gridDim.x;
  const int xcd = bid & 7, slot = bid >> 3, bpx = nb >> 3;
  const long xlo = (long)xcd * n_tiles / 8, xhi = (long)(xcd + 1) * n_tiles / 8;
  // Interleaved assignment inside the XCD's run: at any moment the resident blocks of one XCD work on
  // bpx CONSECUTIVE logical tiles (= a few M-tiles x all their N-tiles), so the A tiles they share and the
  // weight panel stay in that XCD's 4 MiB L2 (a contiguous run per block re-fetched A once per N-tile).
  const int t_begin = (int)xlo + slot, t_end = (int)xhi, t_step = bpx;
  if (t_begin >= t_end) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int c4 = tid & 7, r0 = tid >> 3;              // A staging: float4 column / first row (rows r0 + 32 i)
  const int bq = tid & 3, br = tid >> 2;              // B staging: 8-element k segment / first row (rows br + 64 i)

  const int nk = a.K / BK;
  const int total = GEN ? a.taps * nk : nk;
  const long wplane = a.wx_plane;

  int ld_tile = t_begin, ld_it = 0;
  const float* ap[AI];
  const unsigned short* bp[BI];
  int a_l[GEN ? AI : 1]; long a_m[GEN ? AI : 1];
  const int Lin = GEN ? a.L * a.a_stride : 0;
  auto setup_rows = [&](int tile) {
    const int tile_m = tile / tiles_n;
    const int n0 = (tile - tile_m * tiles_n) * BN;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      int m = tile_m * BM + r0 + 32 * i;
      if (!GEN) {
        m = m < a.M ? m : a.M - 1;
        ap[i] = a.A + (long)m * a.lda + c4 * 4;
      } else {
        const bool mv = m < a.M;
        m = mv ? m : 0;
        const int seg = m / a.L, l = m - seg * a.L;
        a_l[i] = mv ? l * a.a_stride : -(1 << 28);
        a_m[i] = (long)seg * Lin + l * a.a_stride;
      }
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      int n = n0 + br + 64 * i;
      n = n < a.N ? n : a.N - 1;
      bp[i] = a.Wx + (long)n * a.K + bq * 8;
    }
  };
  f32x4 ra[AI];
  u32x4 rb[3][BI];
  auto load_tile = [&]() {
    int k0; long woff;
    if (!GEN) {
      k0 = ld_it * BK; woff = k0;
#pragma unroll
      for (int i = 0; i < AI; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap[i] + k0);
    } else {
      const int tap = ld_it / nk;
      k0 = (ld_it - tap * nk) * BK;
      const int sh = a.shift0 + tap * a.shift_step;
      const float* Ab; int ld, kc;
      if (k0 < a.K1) { Ab = a.A; ld = a.lda; kc = k0; } else { Ab = a.A2; ld = a.lda2; kc = k0 - a.K1; }
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const int l = a_l[i] + sh;
        const bool ok = l >= 0 && l < Lin;
        const long row = ok ? a_m[i] + sh : a_m[i];
        const f32x4 v = *reinterpret_cast<const f32x4*>(Ab + row * ld + kc + c4 * 4);
        ra[i] = ok ? v : f32x4{0, 0, 0, 0};
      }
      woff = (long)tap * a.N * a.K + k0;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < BI; ++i) rb[p][i] = *reinterpret_cast<const u32x4*>(bp[i] + p * wplane + woff);
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      u32x2 p1, p2, p3;
      split3(ra[i], p1, p2, p3);
      const int rowa = r0 + 32 * i;
      const int off = rowa * XLD + ((((c4 >> 1) ^ (rowa >> 2)) & 3) << 3) + (c4 & 1) * 4;
      *reinterpret_cast<u32x2*>(Ax + off) = p1;
      *reinterpret_cast<u32x2*>(Ax + PLANE + off) = p2;
      *reinterpret_cast<u32x2*>(Ax + 2 * PLANE + off) = p3;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < BI; ++i)
        *reinterpret_cast<u32x4*>(Bx + p * PLANE + (br + 64 * i) * XLD + (((bq ^ ((br + 64 * i) >> 2)) & 3) << 3)) = rb[p][i];
  };
  auto advance_loader = [&]() {
    if (++ld_it == total) {
      ld_it = 0;
      if (ld_tile + t_step < t_end) { ld_tile += t_step; setup_rows(ld_tile); }
    }
  };

  const int r = lane & 31, h = lane >> 5;
  // fragment of k-step s2: logical slot 2 s2 + h of row (base + 32 mi + r); (row >> 2) & 3 == (r >> 2) & 3
  const int arow = (wm * 64 + r) * XLD;
  const int brow = (wn * 64 + r) * XLD;
  const int sw = (r >> 2) & 3;

  setup_rows(ld_tile);
  load_tile();
  for (int tile = t_begin; tile < t_end; tile += t_step) {
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    for (int it = 0; it < total; ++it) {
      __syncthreads();                   // readers of the previous slab (or of the C tile) are done
      store_tile();                      // split + registers -> LDS
      __syncthreads();
      advance_loader();
      load_tile();                       // next slab -> registers, in flight during the MFMAs below
      __builtin_amdgcn_sched_barrier(0); // keep the global loads AHEAD of the MFMA block (hipcc sinks them to its end)
      bf16x8 av[2][3][MI], bv[2][3][NI];          // all fragments of the slab first: one exposed LDS latency
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            av[s2][p][mi] = *reinterpret_cast<const bf16x8*>(Ax + p * PLANE + arow + mi * 32 * XLD + (((2 * s2 + h) ^ sw) << 3));
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            bv[s2][p][ni] = *reinterpret_cast<const bf16x8*>(Bx + p * PLANE + brow + ni * 32 * XLD + (((2 * s2 + h) ^ sw) << 3));
        }
      // small terms first; consecutive MFMAs hit different accumulators
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[s2][PA[term]][mi], bv[s2][PB[term]][ni], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();                     // every wave is done with the operand planes
    epilogue<WM, WN, MI, NI, EPI, GEN>(a, acc, smem, tile, tiles_n, tid, wm, wn, r, h);
  }
}

// -------------------------------------------------------------------------------------------------
// bf16x6, software-pipelined: the weight fragments never touch LDS.
//
// The weight planes are pre-packed in MFMA-fragment order ([row/32][k/16][plane][lane][8 bf16], launch_pack_x6),
// so one wave fetches the B operand of a 32x32x16 step with a single fully coalesced 16-byte-per-lane global load
// straight into the registers the MFMA reads (the two waves that share the columns hit the same L1/L2 lines).
// Only the activations go through LDS (they are split into their three planes once, by the staging thread):
// two 24 KB buffers, ONE barrier per 32-wide K slab in the middle of the slab:
//   step 0 of slab s : MFMAs(s, k16 0)  ||  W(s, k16 1) -> regs, A frags (s, k16 1) <- LDS, split + store slab s+1 -> other buffer
//   barrier
//   step 1 of slab s : MFMAs(s, k16 1)  ||  W(s+1, k16 0) -> regs, A frags (s+1, k16 0) <- LDS, global A loads of slab s+2
// so every global / LDS access of a wave is issued one k16 step (24 MFMAs) before its consumer.
// Wave-private epilogue of the pipelined kernels: every wave transposes its own 64 x 64 quadrant through 8.5 KB of LDS
// that only it touches (two passes of 32 rows), so the block needs no barrier inside the epilogue and all four waves
// work at once; the rows still leave as 16-byte stores (a store instruction covers 4 rows x 256 contiguous bytes).
// GEGLU forward expects the weights tiled [32 a-rows | 32 g-rows] so that a wave's two 32-column halves hold matching
// (a, g) pairs.  Same fused math as `epilogue` (bias, per-row-variant bias, residuals, GEGLU forward / backward).
// AUX (EPI_LINEAR): 0 = no addends besides the bias, 1 = one residual, 2 = residual(s) and / or the per-row-variant bias.
// A template parameter rather than run-time `if (a.resid)` tests inside the passes: hipcc places the s_waitcnt of a
// conditional load AFTER the join, where it executes on both paths -- a plain GEMM then waited vmcnt(0) in its second
// pass for loads it never issued, i.e. for the first pass's eight stores to be acknowledged (once per tile).
template <int EPI, bool GEN, bool OSC, int BM, int BN, int AUX, int EABL = 0>   // EABL (diagnostic twin only): 1 no stores, 2 no LDS transpose
__device__ __forceinline__ void epilogue_wave_aux(const GemmArgs& a, f32x16 (&acc)[2][2], float* Sw, int tile, int tiles_n,
                                                  int lane, int wm, int wn, int r, int h, const float oscale) {
  constexpr int SLD = 68;                                  // floats per scratch row (64 + pad: conflict-free b128 reads)
  const int tile_m = tile / tiles_n;
  const int n0 = (tile - tile_m * tiles_n) * BN;
  // Every global LOAD of the epilogue is issued before its first STORE: vmcnt retires in order, so a load behind a store
  // cannot be waited for before that store is acknowledged (and hipcc places the wait after the `if (a.bias)` join, where
  // it executes whether or not anything was loaded).  The biases are therefore fetched once, ahead of both passes.
  f32x4 ba = {0, 0, 0, 0}, bg = {0, 0, 0, 0}, b4 = {0, 0, 0, 0};
  if (EPI == EPI_GEGLU_FWD) {
    const int cg = (lane & 7) * 4, nbg = n0 + wn * 64;
    if (a.bias) { ba = *reinterpret_cast<const f32x4*>(a.bias + nbg + cg); bg = *reinterpret_cast<const f32x4*>(a.bias + nbg + 32 + cg); }
  } else {
    const int nl = n0 + wn * 64 + (lane & 15) * 4;
    if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + (nl < a.N ? nl : 0));
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int m0 = tile_m * BM + wm * 64 + mi * 32;
    if (EPI == EPI_GEGLU_FWD) {
      const int rl0 = lane >> 3, c = (lane & 7) * 4;       // 4 passes of 8 rows; 8 lanes per row
      const int half = a.N >> 1, j = (n0 >> 1) + wn * 32 + c;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          Sw[((reg & 3) + 8 * (reg >> 2) + 4 * h) * SLD + ni * 32 + r] = acc[mi][ni][reg];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int rl = p * 8 + rl0, m = m0 + rl;
        f32x4 av = *reinterpret_cast<const f32x4*>(Sw + rl * SLD + c);
        f32x4 gv = *reinterpret_cast<const f32x4*>(Sw + rl * SLD + 32 + c);
        if (OSC) { av *= oscale; gv *= oscale; }
        av += ba; gv += bg;
        f32x4 hv, s1, s2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float cdf, pdf;
          normal_cdf_pdf(gv[e], cdf, pdf);
          s1[e] = gv[e] * cdf;                           // gelu(g)
          s2[e] = av[e] * (cdf + gv[e] * pdf);           // a * gelu'(g)
          hv[e] = av[e] * s1[e];                         // hg = a * gelu(g)
        }
        if (m < a.M) {
          *reinterpret_cast<f32x4*>(a.C + (long)m * a.ldc + j) = s1;            // stash for the VJP: no transcendental
          *reinterpret_cast<f32x4*>(a.C + (long)m * a.ldc + half + j) = s2;     // is needed in the backward epilogue
          *reinterpret_cast<f32x4*>(a.aux_out + (long)m * a.ld_aux + j) = hv;
        }
      }
    } else {
      const int rl0 = lane >> 4, c = (lane & 15) * 4;      // 8 passes of 4 rows; 16 lanes per row
      const int n = n0 + wn * 64 + c;
      const bool nok = n < a.N;
      const int nc = nok ? n : 0;
      // the global operands of the fused math are requested before the transpose (their latency hides behind it)
      f32x4 pre0[8], pre1[EPI == EPI_GEGLU_BWD ? 8 : 1];
      f32x4 rbv[AUX == 2 && !GEN ? 8 : 1];          // (no conv layer has a per-row-variant bias)
      if (EPI == EPI_LINEAR) {
        if (AUX >= 1 && a.resid) {
#pragma unroll
          for (int p = 0; p < 8; ++p) {
            const int m = m0 + p * 4 + rl0, mc = m < a.M ? m : 0;
            const long orow = GEN ? (long)mc * a.c_rstride + a.c_roff : (long)mc;
            pre0[p] = *reinterpret_cast<const f32x4*>(a.resid + orow * a.ldr + nc);
          }
        }
        if (AUX == 2 && !GEN && a.rowbias) {              // index, then row: both ahead of this pass's stores
          int rvi[8];
#pragma unroll
          for (int p = 0; p < 8; ++p) {
            const int m = m0 + p * 4 + rl0, mc = m < a.M ? m : 0;
            rvi[p] = a.rowvar[a.row0 + mc / a.L];
          }
#pragma unroll
          for (int p = 0; p < 8; ++p) rbv[p] = *reinterpret_cast<const f32x4*>(a.rowbias + (long)rvi[p] * a.rb_stride + nc);
        }
      } else {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int m = m0 + p * 4 + rl0, mc = m < a.M ? m : 0;
          pre0[p] = *reinterpret_cast<const f32x4*>(a.aux_in + (long)mc * a.ld_aux + nc);
          pre1[p] = *reinterpret_cast<const f32x4*>(a.aux_in + (long)mc * a.ld_aux + a.N + nc);
        }
      }
      if (EABL != 2) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          Sw[((reg & 3) + 8 * (reg >> 2) + 4 * h) * SLD + ni * 32 + r] = acc[mi][ni][reg];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int rl = p * 4 + rl0, m = m0 + rl;
        const bool ok = EABL != 1 && nok && m < a.M;
        const int mc = m < a.M ? m : 0;
        const long orow = GEN ? (long)mc * a.c_rstride + a.c_roff : (long)mc;
        f32x4 v;
        if (EABL == 2) v = f32x4{acc[mi][p & 1][(p >> 1) * 4], acc[mi][p & 1][(p >> 1) * 4 + 1], acc[mi][p & 1][(p >> 1) * 4 + 2], acc[mi][p & 1][(p >> 1) * 4 + 3]};
        else v = *reinterpret_cast<const f32x4*>(Sw + rl * SLD + c);
        if (EABL == 1) asm volatile("" :: "v"(v));
        if (OSC) v *= oscale;                            // undo the operand scales (powers of two: exact)
        v += b4;
        if (EPI == EPI_LINEAR) {
          if (AUX == 2 && !GEN && a.rowbias) v += rbv[p];
          if (AUX >= 1 && a.resid) v += pre0[p];
          if (AUX == 2 && GEN && a.resid2) v += *reinterpret_cast<const f32x4*>(a.resid2 + orow * a.ldr2 + nc);
          if (ok) {
            if (!GEN || n < a.N1) *reinterpret_cast<f32x4*>(a.C + orow * a.ldc + n) = v;
            else *reinterpret_cast<f32x4*>(a.C2 + orow * a.ldc2 + (n - a.N1)) = v;
          }
        } else {
          const f32x4 da = v * pre0[p], dg = v * pre1[p];
          if (ok) {
            *reinterpret_cast<f32x4*>(a.C + (long)m * a.ldc + n) = da;
            *reinterpret_cast<f32x4*>(a.C + (long)m * a.ldc + a.N + n) = dg;
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the scratch is rewritten by the next pass
    __builtin_amdgcn_wave_barrier();
  }
}

template <int EPI, bool GEN, bool OSC, int BM = 128, int BN = 128, bool NOAUX = false, int EABL = 0>
__device__ __forceinline__ void epilogue_wave(const GemmArgs& a, f32x16 (&acc)[2][2], float* Sw, int tile, int tiles_n,
                                              int lane, int wm, int wn, int r, int h, const float oscale) {
  if (EABL) { epilogue_wave_aux<EPI, GEN, OSC, BM, BN, 0, EABL>(a, acc, Sw, tile, tiles_n, lane, wm, wn, r, h, oscale); return; }
  if (EPI != EPI_LINEAR || NOAUX) { epilogue_wave_aux<EPI, GEN, OSC, BM, BN, 0>(a, acc, Sw, tile, tiles_n, lane, wm, wn, r, h, oscale); return; }
  if (a.rowbias || (GEN && a.resid2)) epilogue_wave_aux<EPI, GEN, OSC, BM, BN, 2>(a, acc, Sw, tile, tiles_n, lane, wm, wn, r, h, oscale);
  else if (a.resid) epilogue_wave_aux<EPI, GEN, OSC, BM, BN, 1>(a, acc, Sw, tile, tiles_n, lane, wm, wn, r, h, oscale);
  else epilogue_wave_aux<EPI, GEN, OSC, BM, BN, 0>(a, acc, Sw, tile, tiles_n, lane, wm, wn, r, h, oscale);
}

// AMUL: A_eff[m][k] = A[m][k % a_period] * Amul[m][k] (the GEGLU VJP d(ag) = [d(hg) s1 | d(hg) s2] formed while the
// operand is staged, instead of a 2048-wide d(ag) round trip through HBM).
// NP = 3: bf16x6 (three 8-bit planes, six products).  NP = 2: fp16x3 (two 11-bit planes h1 + h2 = 22 significand bits,
// products h1h1' + h1h2' + h2h1'; the dropped h2h2' and the plane residuals are <= 3 * 2^-22 |ab|): half the matrix
// work, but fp16's 5 exponent bits need operands in [2^-14, 2^15] -- see launch_gemm for where it is allowed.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
template <int NP>
__device__ __forceinline__ f32x16 mfma_planes(const u32x4 a, const u32x4 b, const f32x16 c) {
  if (NP == 3) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
// x * s with plain v_mul_f32 (hipcc SLP-packs adjacent multiplies into v_pk_mul_f32, which costs ~13 extra cycles beside MFMAs)
__device__ __forceinline__ void scale4(f32x4& v, float sc) {
  float r0, r1, r2, r3;
  asm("v_mul_f32 %0, %1, %2" : "=v"(r0) : "v"(v[0]), "v"(sc));
  asm("v_mul_f32 %0, %1, %2" : "=v"(r1) : "v"(v[1]), "v"(sc));
  asm("v_mul_f32 %0, %1, %2" : "=v"(r2) : "v"(v[2]), "v"(sc));
  asm("v_mul_f32 %0, %1, %2" : "=v"(r3) : "v"(v[3]), "v"(sc));
  v = f32x4{r0, r1, r2, r3};
}
// one plane of four floats: two packed 16-bit pairs, and (optionally) the floats with that plane removed
template <int NP>
__device__ __forceinline__ u32x2 peel4(f32x4& v, bool subtract) {
  u32x2 q;
  if (NP == 3) {
    q[0] = pk_bf16(v[0], v[1]); q[1] = pk_bf16(v[2], v[3]);
    if (subtract) {
      v[0] -= __builtin_bit_cast(float, q[0] << 16); v[1] -= __builtin_bit_cast(float, q[0] & 0xffff0000u);
      v[2] -= __builtin_bit_cast(float, q[1] << 16); v[3] -= __builtin_bit_cast(float, q[1] & 0xffff0000u);
    }
  } else {
    const half2v h0 = __builtin_convertvector(f32x2{v[0], v[1]}, half2v), h1 = __builtin_convertvector(f32x2{v[2], v[3]}, half2v);
    q[0] = __builtin_bit_cast(unsigned, h0); q[1] = __builtin_bit_cast(unsigned, h1);
    if (subtract) {     // residual in ONE mixed-precision FMA per element: x - h read straight from the packed fp16 half
      float r0, r1, r2, r3;
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(q[0]), "v"(v[0]));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(q[0]), "v"(v[1]));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(q[1]), "v"(v[2]));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(q[1]), "v"(v[3]));
      v = f32x4{r0, r1, r2, r3};
    }
  }
  return q;
}
// REC: the launch also records max |A_eff| (atomic max into a.a_absmax_out) -- the fp16 planes of the NEXT evaluation of
// the same call site are scaled from it (delayed scaling: A * 2^(5 - floor(log2 max)) lands in [2^5, 2^6): the operand
// may grow 2^9.9-fold between two evaluations before fp16 overflows, elements down to 2^-8 of the maximum keep all
// 22 bits and smaller ones an absolute error of 2^-30 of the maximum); a launch whose scaled operand reaches 60000,
// or whose largest scaled element falls below 2^-3, raises a.range_flag.  All scales are powers of two and are undone exactly in the epilogue.
// Two blocks per CU (bf16x6, and the fp16x3 A-multiplier variant whose extra operand registers do not fit three) ...
template <int EPI, bool GEN, bool AMUL = false, int NP = 3, bool REC = false, bool WIDE = false>
__global__ __launch_bounds__(256)
void gemm_x6p_kernel(GemmArgs a, int tiles_n, int n_tiles) {
#define X6P_THREE 0
#define X6P_ABL 0
#define X6P_EABL 0
#define X6P_DEEP 1
#include "gemm_x6p_body.inc"
#undef X6P_EABL
#undef X6P_DEEP
#undef X6P_ABL
#undef X6P_THREE
}
// diagnostic twin of the plain fp16x3 kernel with parts of the loop switched off (ramp_bench_gemm, flags bits 8..11):
// where a tile's time goes.  Results are meaningless; never launched by the product.
template <int ABLV, bool WIDE>
__global__ __launch_bounds__(256)
void gemm_x6p_abl_kernel(GemmArgs a, int tiles_n, int n_tiles) {
  constexpr int EPI = EPI_LINEAR; constexpr bool GEN = false, AMUL = false, REC = true; constexpr int NP = 2;
#define X6P_THREE 0
#define X6P_ABL (ABLV & 15)
#define X6P_EABL ((ABLV >> 5) & 3)
#define X6P_DEEP (!(ABLV & 16))
#define X6P_NOMFMA ((ABLV >> 7) & 1)
#define X6P_TRICKLE (ABLV >> 8)
#include "gemm_x6p_body.inc"
#undef X6P_TRICKLE
#undef X6P_NOMFMA
#undef X6P_EABL
#undef X6P_DEEP
#undef X6P_ABL
#undef X6P_THREE
}
// ... or three (fp16x3: 168 VGPRs, 51 KB of LDS): a third resident block covers the epilogue-store stalls of the others;
// the compiler spills registers around the epilogue (once per tile), none inside the slab loop.  Used where it
// measured faster (launch_x6).
template <int EPI, bool GEN, bool AMUL = false, int NP = 2, bool REC = true, bool WIDE = false, bool PLAIN = false>   // PLAIN: no addend besides the bias
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void gemm_x6p3_kernel(GemmArgs a, int tiles_n, int n_tiles) {
#define X6P_PLAIN PLAIN
#define X6P_THREE 1
#define X6P_ABL 0
#define X6P_EABL 0
#define X6P_DEEP 0
#include "gemm_x6p_body.inc"
#undef X6P_EABL
#undef X6P_DEEP
#undef X6P_ABL
#undef X6P_PLAIN
#undef X6P_THREE
}

// =================================================================================================================
// Fused feed-forward, forward direction (layers_attention_mini.py:38-45, 130-149):
//     z2 = z1 + W2 (a * gelu(g)) + b2 ,   [a | g] = W1 LN3(z1) + b1
// as ONE kernel that never writes the 1024-wide hidden hg = a * gelu(g) to HBM (it is 2048 of the 4864 floats per token
// the FF1 / FF2 pair moves; both layers are K = 256-ish, output-store-bound shapes, profiles/r02_gemm_ablation.txt).
//
// Eight waves per block, one block per CU, an M-tile of 64 tokens at a time, TWO ROLES:
//   G1 = waves 0..3: the register-staged fp16x3 loop of gemm_x6p_body.inc (64 x 256 tile: each wave 64 rows x [32 a | 32 g]
//        packed columns) over one 256-column chunk of W1 at a time (8 chunks per M-tile), and nothing else: at the end of a
//        chunk it drops its raw accumulators into LDS (RAW, 17 KB per wave) and starts the next chunk;
//   G2 = waves 4..7 work on the PREVIOUS chunk while G1 computes: during G1's first four slabs wave w turns G1 wave w's raw
//        tile into GEGLU -- stores the VJP stash [gelu(g) | a gelu'(g)] and writes its 64 x 32 slice of hg, already scaled and
//        split into the two fp16 planes an A operand needs, into the LDS image PH (4 slabs of 32 k, the loop's own layout);
//        during the last four slabs all four run FF2 on PH: 8 k16 steps (A fragments from PH, W2 fragments straight from
//        the packed planes) into a 64 x 256 accumulator that lives across the 8 chunks; after the last chunk bias + residual +
//        store.  The GEGLU math and every global store of the pair therefore sit on waves whose stalls do not hold the
//        FF1 loop (measured on the first version, where G1 carried the GEGLU epilogue: 38 % of the kernel).
// Waves w and w + 4 share a SIMD, so every matrix pipe alternates between one G1 and one G2 wave, and each wave carries
// ONE accumulator set (64 registers): a single-role kernel would need both (128) on top of the loop's ~130 registers.
// Both roles run the SAME loop skeleton -- per chunk interval eight iterations with one block barrier each (G1: one K slab,
// G2: 16 GEGLU rows or two k16 steps) and one hand-off barrier -- so the barrier counts match by construction.
// =================================================================================================================
struct FfFwdArgs {
  GemmArgs g1;     // FF1 as launch_gemm would get it: A = LN3 output (M, 256), Wx = packed [32 a | 32 g]-tiled W1 planes,
                   // bias = packed b1, C = stash (M, 2048), scale / recording slots of ITS call site
  GemmArgs g2;     // FF2: Wx = packed W2 planes (N = 256, K = 1024), bias = b2, resid = z1, C = z2; a_absmax_in / _out /
                   // range_flag / site_id / w_scale_inv of ITS call site (the operand is hg)
};

constexpr int FF_PLANE = 64 * XLD;                        // halfs per plane of a 64-row, 32-k slab image
constexpr int FF_BUF = 2 * FF_PLANE;                      // halfs per slab image (two planes): 8 KB
constexpr int FF_PH = 4 * FF_BUF * 2;                     // bytes of the hidden-chunk image (4 slabs): 32 KB
constexpr int FF_SCR = 32 * 68 * 4;                       // bytes of one wave scratch of the FF2 epilogue (32 rows x 64 columns + pad)
constexpr int FF_RAW = 64 * 68 * 4;                       // bytes of one G1 wave's raw [a | g] tile (64 rows x 64 columns + pad)
constexpr size_t FF_LDS = 2 * (size_t)FF_BUF * 2 + 4 * (size_t)FF_RAW + (size_t)FF_PH + 4 * (size_t)FF_SCR;   // 16 + 68 + 32 + 34 KB
static_assert(FF_LDS <= 160 * 1024, "LDS budget of the fused feed-forward kernel");

__global__ __launch_bounds__(512)
void ff_fwd_kernel(FfFwdArgs f, int n_mtiles) {
  constexpr int NP = 2, MI = 2, NI = 2, AI = 2, NMF = 12;
  constexpr int BUF = FF_BUF, PLANE = FF_PLANE;
  constexpr long WBLK = NP * 1024;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short* Ax = reinterpret_cast<unsigned short*>(smem);                                   // [2][2][64][XLD]
  float* RAW = smem + (2 * BUF * 2) / 4;                                                          // [4 waves][64][68]
  unsigned short* PH = reinterpret_cast<unsigned short*>(RAW + 4 * (FF_RAW / 4));                 // [4][2][64][XLD]
  const GemmArgs& a = f.g1;
  const GemmArgs& b = f.g2;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool G1 = wave < 4;
  const int wn = wave & 3;
  float* Sw = smem + (2 * BUF * 2 + 4 * FF_RAW + FF_PH) / 4 + (wave & 3) * (FF_SCR / 4);       // FF2 epilogue scratch (G2 waves)

  // M-tiles of this block: its XCD's contiguous run, interleaved over the XCD's blocks
  const int bid = blockIdx.x, nb = gridDim.x;
  const int xcd = bid & 7, slot = bid >> 3, bpx = nb >> 3;
  const long xlo = (long)xcd * n_mtiles / 8, xhi = (long)(xcd + 1) * n_mtiles / 8;
  const int mt_begin = (int)xlo + slot, mt_end = (int)xhi, mt_step = bpx;
  if (mt_begin >= mt_end) return;
  const int n_my = (mt_end - mt_begin + mt_step - 1) / mt_step;
  const int Q = n_my * 8;                                   // chunks this block processes

  auto scale_of = [](const float* p) {
    float s = 1.f;
    const float mx = p ? *p : 0.f;
    if (mx > 0.f) {
      int eb = (int)((__builtin_bit_cast(unsigned, mx) >> 23) & 0xffu);
      int sb = 259 - eb;
      sb = sb < 1 ? 1 : (sb > 254 ? 254 : sb);
      s = __builtin_bit_cast(float, (unsigned)sb << 23);
    }
    return s;
  };
  const float s_a = scale_of(a.a_absmax_in), s_h = scale_of(b.a_absmax_in);
  const float oscale1 = a.w_scale_inv / s_a, oscale2 = b.w_scale_inv / s_h;
  float amax = 0.f, amax_h = 0.f;

  const int r = lane & 31, h = lane >> 5;
  const int arow = r * XLD;
  const int sw = (r >> 2) & 3;
  const unsigned wlane = lane * 16;

  // ------------------------------------------------------------------------------------------------------------------
  // G1 state: the loader (one slab ahead of the LDS stores, two ahead in registers, across chunk and tile boundaries)
  // ------------------------------------------------------------------------------------------------------------------
  const int tg = tid & 255;
  const int c4 = tg & 7, r0 = tg >> 3;
  int ld_q = 0, ld_it = 0;
  const char* abase = nullptr; unsigned aoff[AI];
  auto setup_rows = [&](int mt) {
    abase = reinterpret_cast<const char*>(a.A + (long)mt * 64 * a.lda);
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      int m = mt * 64 + r0 + 32 * i;
      m = m < a.M ? m : a.M - 1;
      aoff[i] = (unsigned)(m - mt * 64) * (unsigned)a.lda * 4u + c4 * 16u;
    }
  };
  f32x4 ra[AI];
  auto load_tile = [&]() {
    const int k0 = ld_it * BK;
#pragma unroll
    for (int i = 0; i < AI; ++i) ra[i] = *reinterpret_cast<const f32x4*>(abase + (long)k0 * 4 + aoff[i]);
  };
  auto store_tile = [&](unsigned short* dst) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      f32x4 v = ra[i];
      amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), amax); amax = fmaxf(fmaxf(fabsf(v[2]), fabsf(v[3])), amax);
      scale4(v, s_a);
      const int rowa = r0 + 32 * i;
      const int off = rowa * XLD + ((((c4 >> 1) ^ (rowa >> 2)) & 3) << 3) + (c4 & 1) * 4;
#pragma unroll
      for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(dst + p * PLANE + off) = peel4<NP>(v, p + 1 < NP);
    }
  };
  auto advance_loader = [&]() {
    if (++ld_it == 8) {
      ld_it = 0;
      if (ld_q + 1 < Q) { ++ld_q; if ((ld_q & 7) == 0) setup_rows(mt_begin + (ld_q >> 3) * mt_step); }
    }
  };
  const char* wl = reinterpret_cast<const char*>(a.Wx);
  auto w_blocks = [&](int q, long (&blk)[NI]) {             // W1 fragment blocks of this wave's two 32-column groups in chunk q & 7
    const int n0 = (q & 7) * 256 + wn * 64;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) blk[ni] = (long)((n0 + ni * 32) >> 5) * 16;       // K = 256: 16 k16 blocks per group
  };
  u32x4 av[2][NP][MI], bw[2][NP][NI];
  long wblk[NI], wnext[NI];
  auto load_w = [&](u32x4 (&dst)[NP][NI], const long (&blk)[NI], int it, int s2) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const char* q = wl + (blk[ni] + 2 * it + s2) * WBLK;
#pragma unroll
      for (int p = 0; p < NP; ++p) dst[p][ni] = *reinterpret_cast<const u32x4*>(q + p * 1024 + wlane);
    }
  };
  auto read_a = [&](u32x4 (&dst)[NP][MI], const unsigned short* src, int s2) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        dst[p][mi] = *reinterpret_cast<const u32x4*>(src + p * PLANE + arow + mi * 32 * XLD + (((2 * s2 + h) ^ sw) << 3));
  };
  constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};       // small terms first: h2 h1', h1 h2', h1 h1'
  int parity = 0;

  // ------------------------------------------------------------------------------------------------------------------
  // G2 state
  // ------------------------------------------------------------------------------------------------------------------
  const char* wl2 = reinterpret_cast<const char*>(b.Wx);
  long blk2[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) blk2[ni] = (long)(wn * 2 + ni) * 64;                 // K = 1024: 64 k16 blocks per 32-column group
  auto load_w2 = [&](u32x4 (&dst)[NP][NI], int kb) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int p = 0; p < NP; ++p) dst[p][ni] = *reinterpret_cast<const u32x4*>(wl2 + (blk2[ni] + kb) * WBLK + p * 1024 + wlane);
  };

  // The two roles are two separate loop nests with the SAME barrier structure (1 prologue barrier; per chunk interval
  // q = 0 .. Q: eight slab barriers + one hand-off barrier): one nest with role tests inside would make every
  // loop-carried value of either role live in both (the FF2 accumulator through G1's epilogue, ...) and spill.
  if (G1) {
    if (!(a.ablate & 16)) __builtin_amdgcn_s_setprio(2);     // the FF1 loop is the critical path: its wave wins the issue arbitration
    // ---- prologue --------------------------------------------------------------------------------------------------
    setup_rows(mt_begin);
    load_tile();
    w_blocks(0, wblk);
    load_w(bw[0], wblk, 0, 0);
    load_w(bw[1], wblk, 0, 1);
    store_tile(Ax);
    advance_loader();
    load_tile();
    __syncthreads();
    read_a(av[0], Ax, 0);
    for (int q = 0; q <= Q; ++q) {
      if (q == Q) {                                          // G2 finishes the last chunk: barriers only
#pragma unroll 1
        for (int it = 0; it < 9; ++it) __syncthreads();
        break;
      }
      f32x16 acc[MI][NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
      if (q + 1 < Q) w_blocks(q + 1, wnext);
      else {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wnext[ni] = wblk[ni];
      }
#pragma unroll 1
      for (int it = 0; it < 8; ++it) {
        advance_loader();
        unsigned short* cur = Ax + parity * BUF;
        unsigned short* oth = Ax + (parity ^ 1) * BUF;
        {
          const long kb = 2 * it + 1;
#pragma unroll
          for (int j = 0; j < NMF; ++j) {
            acc[(j >> 1) & 1][j & 1] = mfma_planes<NP>(av[0][PA[j >> 2]][(j >> 1) & 1], bw[0][PB[j >> 2]][j & 1], acc[(j >> 1) & 1][j & 1]);
            if (j < 2 * NP) {
              const int p = j >> 1, x = j & 1;
              if (it != 0) bw[1][p][x] = *reinterpret_cast<const u32x4*>(wl + (wblk[x] + kb) * WBLK + p * 1024 + wlane);
              av[1][p][x] = *reinterpret_cast<const u32x4*>(cur + p * PLANE + arow + x * 32 * XLD + (((2 + h) ^ sw) << 3));
            } else if (j < (2 + AI) * NP) {
              const int i = (j - 2 * NP) / NP, st = (j - 2 * NP) % NP;
              if (st == 0) {
                amax = fmaxf(fmaxf(fabsf(ra[i][0]), fabsf(ra[i][1])), amax);
                amax = fmaxf(fmaxf(fabsf(ra[i][2]), fabsf(ra[i][3])), amax);
                scale4(ra[i], s_a);
              }
              const int rowa = r0 + 32 * i;
              const int off = rowa * XLD + ((((c4 >> 1) ^ (rowa >> 2)) & 3) << 3) + (c4 & 1) * 4;
              *reinterpret_cast<u32x2*>(oth + st * PLANE + off) = peel4<NP>(ra[i], st + 1 < NP);
            }
            if (j == NMF - 1) load_tile();
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __syncthreads();
        {
          const bool last = it == 7;
          const int it_next = last ? 0 : it + 1;
          long wsel[NI];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) wsel[ni] = last ? wnext[ni] : wblk[ni];
          const long kb = 2 * it_next;
#pragma unroll
          for (int j = 0; j < NMF; ++j) {
            acc[(j >> 1) & 1][j & 1] = mfma_planes<NP>(av[1][PA[j >> 2]][(j >> 1) & 1], bw[1][PB[j >> 2]][j & 1], acc[(j >> 1) & 1][j & 1]);
            if (j < 2 * NP) {
              const int p = j >> 1, x = j & 1;
              bw[0][p][x] = *reinterpret_cast<const u32x4*>(wl + (wsel[x] + kb) * WBLK + p * 1024 + wlane);
              av[0][p][x] = *reinterpret_cast<const u32x4*>(oth + p * PLANE + arow + x * 32 * XLD + ((h ^ sw) << 3));
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        parity ^= 1;
      }
      // the next chunk's second-step fragments are requested before the stores of this one (vmcnt retires in order)
      load_w(bw[1], wnext, 0, 1);
      if (a.ablate & 2) {                                    // diagnostic (ramp_bench_gemm): main loop only
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) asm volatile("" :: "v"(acc[qq >> 1][qq & 1]));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wblk[ni] = wnext[ni];
        __syncthreads();
        continue;
      }
      // ---- hand the chunk's raw [a | g] accumulators to G2 (wave-private 64 x 64 image, the C layout transposed) -------
      {
        float* Rw = RAW + wn * (FF_RAW / 4);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
              Rw[(mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h) * 68 + ni * 32 + r] = acc[mi][ni][reg];
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wblk[ni] = wnext[ni];
      __syncthreads();                                        // hand-off: RAW holds chunk q
    }
  } else {
    u32x4 fa[NP][MI], fb[2][NP][NI];
    load_w2(fb[0], 0);
    __syncthreads();                                          // (G1's prologue barrier)
#pragma unroll 1
    for (int it = 0; it < 9; ++it) __syncthreads();           // interval 0: G1 computes the first chunk
    f32x16 acc[MI][NI];
    for (int cq = 0; cq < Q; ++cq) {                          // interval q = cq + 1
      if ((cq & 7) == 0) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
      }
      if (a.ablate & 1) {                                    // diagnostic (ramp_bench_gemm): this role only keeps the barriers
#pragma unroll 1
        for (int it = 0; it < 9; ++it) __syncthreads();
        continue;
      }
      const int mtq = mt_begin + (cq >> 3) * mt_step, cidx = cq & 7;
      // ---- slabs 0..3 of G1's next chunk: GEGLU of wave wn's raw tile, 16 rows per slab ---------------------------------
      {
        const float* Rw = RAW + wn * (FF_RAW / 4);
        const int rl0 = lane >> 3, c = (lane & 7) * 4;       // 8 rows per step; 8 lanes per row
        const int nbase = cidx * 256 + wn * 64;              // this tile's packed columns: [32 a | 32 g]
        const int half = a.N >> 1, j0 = cidx * 128 + wn * 32 + c;
        f32x4 ba = {0, 0, 0, 0}, bg = {0, 0, 0, 0};
        if (a.bias) { ba = *reinterpret_cast<const f32x4*>(a.bias + nbase + c); bg = *reinterpret_cast<const f32x4*>(a.bias + nbase + 32 + c); }
        unsigned short* PHw = PH + wn * BUF;
#pragma unroll 1
        for (int it = 0; it < 4; ++it) {
          if (a.ablate & 8) { __syncthreads(); continue; }   // diagnostic: no GEGLU phase
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const int rl = (it * 2 + p) * 8 + rl0, m = mtq * 64 + rl;
            f32x4 av4 = *reinterpret_cast<const f32x4*>(Rw + rl * 68 + c);
            f32x4 gv = *reinterpret_cast<const f32x4*>(Rw + rl * 68 + 32 + c);
            av4 = av4 * oscale1 + ba; gv = gv * oscale1 + bg;
            f32x4 hv, s1, s2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float cdf, pdf;
              normal_cdf_pdf(gv[e], cdf, pdf);
              s1[e] = gv[e] * cdf;                           // gelu(g)
              s2[e] = av4[e] * (cdf + gv[e] * pdf);          // a * gelu'(g)
              hv[e] = av4[e] * s1[e];                        // hg = a * gelu(g)
            }
            if (m < a.M) {
              *reinterpret_cast<f32x4*>(a.C + (long)m * a.ldc + j0) = s1;
              *reinterpret_cast<f32x4*>(a.C + (long)m * a.ldc + half + j0) = s2;
            } else {
              hv = f32x4{0, 0, 0, 0};                        // rows past M: keep the image finite, never stored
            }
            amax_h = fmaxf(fmaxf(fabsf(hv[0]), fabsf(hv[1])), amax_h); amax_h = fmaxf(fmaxf(fabsf(hv[2]), fabsf(hv[3])), amax_h);
            scale4(hv, s_h);
            const int c8 = lane & 7;
            const int off = rl * XLD + ((((c8 >> 1) ^ (rl >> 2)) & 3) << 3) + (c8 & 1) * 4;
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) *reinterpret_cast<u32x2*>(PHw + pp * PLANE + off) = peel4<NP>(hv, pp + 1 < NP);
          }
          __syncthreads();
        }
      }
      // ---- slabs 4..7: FF2 on the image, two k16 steps (one 32-k slab of it) per slab -----------------------------------
#pragma unroll 1
      for (int it = 0; it < 4; ++it) {
        if (a.ablate & 4) { __syncthreads(); continue; }     // diagnostic: no FF2 phase
        const int kb = cidx * 8 + 2 * it;
        load_w2(fb[1], (kb + 1) & 63);
        read_a(fa, PH + it * BUF, 0);
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi][ni] = mfma_planes<NP>(fa[PA[t3]][mi], fb[0][PB[t3]][ni], acc[mi][ni]);
        load_w2(fb[0], (kb + 2) & 63);                        // (the next chunk's first step at it == 3)
        read_a(fa, PH + it * BUF, 1);
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi][ni] = mfma_planes<NP>(fa[PA[t3]][mi], fb[1][PB[t3]][ni], acc[mi][ni]);
        __syncthreads();
      }
      if ((cq & 7) == 7) {
        // ---- FF2 epilogue of the M-tile: z2 = acc * oscale2 + b2 + z1 -------------------------------------------------
        epilogue_wave_aux<EPI_LINEAR, false, true, 64, 256, 1>(b, acc, Sw, mtq, 1, lane, 0, wn, r, h, oscale2);
      }
      __syncthreads();                                        // hand-off
    }
  }

  // ---- maxima for the next evaluation's scales, range guard (as in the stand-alone kernels): G1 saw LN3's output, G2 saw hg ------
  {
    float mx = G1 ? amax : amax_h;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) {
      const GemmArgs& g = G1 ? a : b;
      const float sc = G1 ? s_a : s_h;
      record_amax(g.a_absmax_out, mx);
      if (g.range_flag && (!(mx * sc < 60000.f) || (mx > 0.f && mx * sc < 0.125f))) atomicMax(g.range_flag, g.site_id + 1);
    }
  }
}

int launch_ff_fwd(const GemmArgs& g1, const GemmArgs& g2, hipStream_t s) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  RAMP_REQUIRE(g1.wx_packed == 2 && g2.wx_packed == 2 && g1.Wx && g2.Wx, "fused feed-forward needs the fp16 weight planes");
  RAMP_REQUIRE(g1.K == 256 && g1.N == 2048 && g2.K == 1024 && g2.N == 256 && g1.M == g2.M && g1.M > 0, "fused feed-forward: 256 -> 2 x 1024 -> 256 only");
  RAMP_REQUIRE(g1.geglu_group == 32 && g1.taps == 1 && g2.taps == 1 && !g1.A2 && !g1.Amul && !g2.rowbias && !g2.resid2 && !g2.C2, "fused feed-forward: unsupported operand");
  RAMP_REQUIRE(g1.lda % 4 == 0 && g1.ldc % 4 == 0 && g2.ldc % 4 == 0 && (g2.resid == nullptr || g2.ldr % 4 == 0), "leading dimensions must be multiples of 4 floats");
  RAMP_REQUIRE(al16(g1.A) && al16(g1.Wx) && al16(g2.Wx) && al16(g1.C) && al16(g2.C) && al16(g1.bias) && al16(g2.bias) && al16(g2.resid), "operands must be 16-byte aligned");
  FfFwdArgs f; f.g1 = g1; f.g2 = g2;
  f.g2.N1 = f.g2.N; f.g2.K1 = f.g2.K; f.g1.N1 = f.g1.N; f.g1.K1 = f.g1.K;
  const int n_mtiles = (g1.M + 63) / 64;
  const int nb = std::min(((n_mtiles + 7) / 8) * 8, 256);   // one 8-wave block per CU
  hipLaunchKernelGGL(ff_fwd_kernel, dim3(nb), dim3(512), FF_LDS, s, f, n_mtiles);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

// fp32 [rows][K] -> fragment-packed planes [rows/32][K/16][NP][64 lanes][8] (NP = 3: bf16, NP = 2: fp16)
template <int NP>
__global__ void pack_planes_kernel(const float* __restrict__ W, unsigned short* __restrict__ out, long rows, int K, float scale) {
  const int k8n = K / 8;
  const long total = rows * k8n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / k8n;
    const int k8 = (int)(i - row * k8n);
    f32x4 lo = *reinterpret_cast<const f32x4*>(W + row * K + k8 * 8) * scale;
    f32x4 hi = *reinterpret_cast<const f32x4*>(W + row * K + k8 * 8 + 4) * scale;
    const long blk = (row >> 5) * (K / 16) + (k8 >> 1);
    unsigned short* q = out + blk * (NP * 512) + (((k8 & 1) * 32 + (int)(row & 31)) << 3);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const bool sub = p + 1 < NP;
      const u32x2 qa = peel4<NP>(lo, sub), qb = peel4<NP>(hi, sub);
      *reinterpret_cast<u32x4*>(q + p * 512) = u32x4{qa[0], qa[1], qb[0], qb[1]};
    }
  }
}
int launch_pack_x6(const float* W, unsigned short* out, long rows, int K, hipStream_t s) {
  RAMP_REQUIRE(rows > 0 && rows % 32 == 0 && K > 0 && K % 16 == 0, "pack_x6 needs rows % 32 == 0 and K % 16 == 0");
  const long total = rows * (K / 8);
  hipLaunchKernelGGL(pack_planes_kernel<3>, dim3((int)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, s, W, out, rows, K, 1.f);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}
int launch_pack_h3(const float* W, unsigned short* out, long rows, int K, float scale, hipStream_t s) {
  RAMP_REQUIRE(rows > 0 && rows % 32 == 0 && K > 0 && K % 16 == 0, "pack_h3 needs rows % 32 == 0 and K % 16 == 0");
  const long total = rows * (K / 8);
  hipLaunchKernelGGL(pack_planes_kernel<2>, dim3((int)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, s, W, out, rows, K, scale);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

constexpr size_t X6_LDS = std::max<size_t>(6 * (size_t)128 * XLD * 2, (size_t)128 * 132 * 4);
constexpr size_t X6P_LDS3 = 2 * (9216 + 3 * (size_t)128 * XLD * 2);    // pipelined, 3 planes: 67584 B (2 blocks / CU)
constexpr size_t X6P_LDS2 = 2 * (9728 + 2 * (size_t)128 * XLD * 2);    // pipelined, 2 planes, 3 blocks / CU: 52224 B
constexpr size_t X6P_LDS2R = 2 * (17408 + 2 * (size_t)128 * XLD * 2);  // pipelined, 2 planes, 2 blocks / CU (roomy): 67584 B
constexpr size_t X6P_LDS2W = 2 * (17408 + 2 * (size_t)64 * XLD * 2);   // 64 x 256 tile, 2 planes (roomy): 51200 B (up to 3 blocks / CU)
static_assert(X6P_LDS3 <= X6_LDS, "bf16x6 pipelined layout");

template <int EPI, bool GEN>
static int launch_x6(const GemmArgs& a, hipStream_t s) {
  // 64 x 256 tiles (fp16x3 linears whose N is a multiple of 256: every transformer projection): see gemm_x6p_body.inc
  const bool wide = !GEN && a.wx_packed == 2 && a.N % 256 == 0 && a.tile_pref != 1;
  const int BMt = wide ? 64 : 128, BNt = wide ? 256 : 128;
  const int tiles_m = (a.M + BMt - 1) / BMt, tiles_n = (a.N + BNt - 1) / BNt;
  const int n_tiles = tiles_m * tiles_n;
  // fp16x3 GEGLU forward (the kernel with the longest epilogue): 3 blocks per CU.  The plain shapes measured slower that
  // way (the ~60 registers spilled around every tile's epilogue cost more than the third block hides: 228 -> 172 TFLOP/s
  // at 393216 x 256 x 256), GEGLU forward faster (189 -> 201).
  // A third resident block where it measured faster (ramp_amd/tools, M = 49152 .. 393216): the bias-only epilogue (no residual /
  // row-variant prefetch registers: 168 VGPRs without a spill) from four tiles per block up, +4..9 %; the A-multiplier kernel
  // only where it turns two rounds into one (+13 %).  Fewer tiles per block lose more to the un-amortised prologue than the
  // third block hides (-1..-40 %).
  const bool three_auto = a.three_ok != 0;                // launch-plan knob (ramp_launch_plan.three_blocks)
  const bool plain = three_auto && EPI == EPI_LINEAR && wide && !a.Amul && !a.resid && !a.rowbias;
  const bool three = a.wx_packed == 2 && (!a.Amul || wide) &&
                     (EPI == EPI_GEGLU_FWD || a.tile_pref == 3 ||
                      (a.tile_pref == 0 && ((plain && n_tiles >= 3072) || (three_auto && wide && a.Amul && n_tiles > 512 && n_tiles <= 768))));
  const int slots = three ? 768 : 512;
  const int rounds = (n_tiles + slots - 1) / slots;
  const int nb = std::min((((n_tiles + rounds - 1) / rounds + 7) / 8) * 8, slots);
#define X6P_LAUNCH(...) hipLaunchKernelGGL((gemm_x6p_kernel<__VA_ARGS__>), dim3(nb), dim3(256), X6P_LDS3, s, a, tiles_n, n_tiles)
  const bool rec = a.a_absmax_out != nullptr;
  if (a.ablate && a.wx_packed == 2) {
    if constexpr (!GEN && EPI == EPI_LINEAR) {
      const size_t lds = wide ? X6P_LDS2W : X6P_LDS2R;
#define ABL_CASE(V) case V: if (wide) hipLaunchKernelGGL((gemm_x6p_abl_kernel<V, true>), dim3(nb), dim3(256), lds, s, a, tiles_n, n_tiles); \
                            else hipLaunchKernelGGL((gemm_x6p_abl_kernel<V, false>), dim3(nb), dim3(256), lds, s, a, tiles_n, n_tiles); break;
      switch (a.ablate) { ABL_CASE(1) ABL_CASE(2) ABL_CASE(3) ABL_CASE(4) ABL_CASE(7) ABL_CASE(8) ABL_CASE(15) ABL_CASE(16) ABL_CASE(32) ABL_CASE(64) ABL_CASE(128) ABL_CASE(136) ABL_CASE(129) ABL_CASE(130) ABL_CASE(131) ABL_CASE(160) ABL_CASE(192) ABL_CASE(144) ABL_CASE(288) ABL_CASE(416) default: RAMP_REQUIRE(false, "ablation variant not built"); }
#undef ABL_CASE
    }
  } else if (wide) {
    if constexpr (!GEN) {
      if (a.Amul) {
        RAMP_REQUIRE(EPI == EPI_LINEAR && a.a_period > 0 && a.a_period % 32 == 0 && a.K == 2 * a.a_period && a.lda_mul % 4 == 0, "bad A-multiplier operand (K = 2 x period)");
        if constexpr (EPI == EPI_LINEAR) {
          if (three) hipLaunchKernelGGL((gemm_x6p3_kernel<EPI_LINEAR, false, true, 2, true, true, true>), dim3(nb), dim3(256), X6P_LDS2W, s, a, tiles_n, n_tiles);
          else hipLaunchKernelGGL((gemm_x6p_kernel<EPI_LINEAR, false, true, 2, true, true>), dim3(nb), dim3(256), X6P_LDS2W, s, a, tiles_n, n_tiles);
        }
      } else if (three && EPI == EPI_LINEAR && !a.resid && !a.rowbias) hipLaunchKernelGGL((gemm_x6p3_kernel<EPI, false, false, 2, true, true, true>), dim3(nb), dim3(256), X6P_LDS2W, s, a, tiles_n, n_tiles);
      else if (three) hipLaunchKernelGGL((gemm_x6p3_kernel<EPI, false, false, 2, true, true>), dim3(nb), dim3(256), X6P_LDS2W, s, a, tiles_n, n_tiles);
      else hipLaunchKernelGGL((gemm_x6p_kernel<EPI, false, false, 2, true, true>), dim3(nb), dim3(256), X6P_LDS2W, s, a, tiles_n, n_tiles);
    }
  } else if (a.wx_packed && a.Amul) {
    RAMP_REQUIRE(EPI == EPI_LINEAR && !GEN && a.a_period > 0 && a.a_period % 32 == 0 && a.K == 2 * a.a_period && a.lda_mul % 4 == 0, "bad A-multiplier operand (K = 2 x period)");
    if (a.wx_packed == 2) hipLaunchKernelGGL((gemm_x6p_kernel<EPI_LINEAR, false, true, 2, true>), dim3(nb), dim3(256), X6P_LDS2R, s, a, tiles_n, n_tiles);
    else if (rec) X6P_LAUNCH(EPI_LINEAR, false, true, 3, true);
    else X6P_LAUNCH(EPI_LINEAR, false, true, 3, false);
  } else if (three) hipLaunchKernelGGL((gemm_x6p3_kernel<EPI, GEN>), dim3(nb), dim3(256), X6P_LDS2, s, a, tiles_n, n_tiles);
  else if (a.wx_packed == 2) hipLaunchKernelGGL((gemm_x6p_kernel<EPI, GEN, false, 2, true>), dim3(nb), dim3(256), X6P_LDS2R, s, a, tiles_n, n_tiles);
  else if (a.wx_packed && rec) X6P_LAUNCH(EPI, GEN, false, 3, true);
  else if (a.wx_packed) X6P_LAUNCH(EPI, GEN, false, 3, false);
  else hipLaunchKernelGGL((gemm_x6_kernel<EPI, GEN>), dim3(nb), dim3(256), X6_LDS, s, a, tiles_n, n_tiles);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}
template <int EPI, bool GEN>
static int set_attr_x6() {
  RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x6_kernel<EPI, GEN>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)X6_LDS));
#define X6P_ATTR(...) RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x6p_kernel<__VA_ARGS__>), \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)X6P_LDS3))
  X6P_ATTR(EPI, GEN, false, 3, false);
  X6P_ATTR(EPI, GEN, false, 3, true);
  RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x6p3_kernel<EPI, GEN>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)X6P_LDS2));
  RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x6p_kernel<EPI, GEN, false, 2, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)X6P_LDS2R));
  if constexpr (!GEN) {
    RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x6p_kernel<EPI, false, false, 2, true, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)X6P_LDS2W));
    RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x6p3_kernel<EPI, false, false, 2, true, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)X6P_LDS2W));
  }
  return 0;
}
template <int WM, int WN, int MI, int NI> struct Cfg {
  static constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
  static constexpr size_t lds = 2 * (size_t)(BM + BN) * LDS_LD * sizeof(float);
};

template <int WM, int WN, int MI, int NI, int EPI, bool GEN>
static int launch_cfg(const GemmArgs& a, int blocks_per_cu, hipStream_t s) {
  using C = Cfg<WM, WN, MI, NI>;
  const int tiles_m = (a.M + C::BM - 1) / C::BM, tiles_n = (a.N + C::BN - 1) / C::BN;
  const int n_tiles = tiles_m * tiles_n;
  // fewest rounds the resident slots allow, then the fewest blocks that still finish in that many rounds
  const int slots = 256 * blocks_per_cu;
  const int rounds = (n_tiles + slots - 1) / slots;
  const int nb = std::min((((n_tiles + rounds - 1) / rounds + 7) / 8) * 8, slots);
  hipLaunchKernelGGL((gemm_kernel<WM, WN, MI, NI, EPI, GEN>), dim3(nb), dim3(WM * WN * 64), C::lds, s, a, tiles_n, n_tiles);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

template <int WM, int WN, int MI, int NI, int EPI, bool GEN>
static int set_attr() {
  RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<WM, WN, MI, NI, EPI, GEN>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg<WM, WN, MI, NI>::lds));
  return 0;
}
int init_gemm_attributes() {
  RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ff_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FF_LDS));
  if (int e = set_attr<2, 2, 2, 2, EPI_LINEAR, false>()) return e;
  if (int e = set_attr<2, 2, 2, 2, EPI_LINEAR, true>()) return e;
  if (int e = set_attr<2, 2, 2, 2, EPI_GEGLU_FWD, false>()) return e;
  if (int e = set_attr<2, 2, 2, 2, EPI_GEGLU_BWD, false>()) return e;
  if (int e = set_attr<2, 2, 2, 1, EPI_LINEAR, true>()) return e;
  if (int e = set_attr<4, 1, 1, 1, EPI_LINEAR, true>()) return e;
  X6P_ATTR(EPI_LINEAR, false, true, 3, false);
  X6P_ATTR(EPI_LINEAR, false, true, 3, true);
  RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x6p_kernel<EPI_LINEAR, false, true, 2, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)X6P_LDS2R));
  RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x6p_kernel<EPI_LINEAR, false, true, 2, true, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)X6P_LDS2W));
#define ABL_ATTR(V) RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x6p_abl_kernel<V, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)X6P_LDS2W)); \
                    RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x6p_abl_kernel<V, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)X6P_LDS2R));
  ABL_ATTR(1) ABL_ATTR(2) ABL_ATTR(3) ABL_ATTR(4) ABL_ATTR(7) ABL_ATTR(8) ABL_ATTR(15) ABL_ATTR(16) ABL_ATTR(32) ABL_ATTR(64)
#undef ABL_ATTR
  if (int e = set_attr_x6<EPI_LINEAR, false>()) return e;
  if (int e = set_attr_x6<EPI_LINEAR, true>()) return e;
  if (int e = set_attr_x6<EPI_GEGLU_FWD, false>()) return e;
  return set_attr_x6<EPI_GEGLU_BWD, false>();
}

int launch_gemm(const GemmArgs& a_in, hipStream_t s) {
  GemmArgs a = a_in;
  if (a.A2 == nullptr) a.K1 = a.K;
  if (a.C2 == nullptr) a.N1 = a.N;
  if (a.rb_stride == 0) a.rb_stride = a.N;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  RAMP_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.taps >= 1 && a.L >= 1, "bad GEMM dims");
  RAMP_REQUIRE(a.K % BK == 0 && a.lda % 4 == 0 && a.N % 4 == 0 && a.N1 % 4 == 0,
               "K must be a multiple of 32, N, N1 and lda multiples of 4 floats");
  RAMP_REQUIRE(a.A2 == nullptr || (a.lda2 % 4 == 0 && a.K1 % BK == 0), "split-K source must start on a K tile");
  RAMP_REQUIRE(al16(a.A) && al16(a.W) && al16(a.A2) && al16(a.C) && al16(a.C2) && al16(a.resid) && al16(a.resid2) &&
               al16(a.bias) && al16(a.rowbias) && al16(a.aux_in) && al16(a.aux_out), "GEMM operands must be 16-byte aligned");
  RAMP_REQUIRE(a.ldc % 4 == 0 && a.ldc2 % 4 == 0 && a.ldr % 4 == 0 && a.ldr2 % 4 == 0 && a.ld_aux % 4 == 0 &&
               a.rb_stride % 4 == 0, "leading dimensions must be multiples of 4 floats");
  RAMP_REQUIRE(a.rowbias == nullptr || a.rowvar != nullptr, "rowbias needs rowvar");
  RAMP_REQUIRE(a.a_stride >= 1 && a.c_rstride >= 1, "bad strides");
  const bool gen = a.taps > 1 || a.a_stride != 1 || a.c_rstride != 1 || a.c_roff != 0 || a.A2 || a.C2 || a.resid2 ||
                   a.shift0 != 0;
  // fragment-packed split-precision weights also serve N = 64 (half of every 128-wide tile is discarded, which still
  // beats the fp32 matrix pipe: the 64-channel 5-tap convolutions were fp32-MFMA-bound at 65-80 TFLOP/s)
  const bool x6 = a.Wx != nullptr && (a.N >= 128 || (a.wx_packed && a.N >= 64));
  RAMP_REQUIRE(a.Amul == nullptr || (x6 && a.wx_packed && !gen && a.epi == EPI_LINEAR && al16(a.Amul)),
               "the A-multiplier operand needs the pipelined bf16x6 kernel");
  RAMP_REQUIRE(!x6 || (al16(a.Wx) && a.K % 8 == 0 && (a.wx_packed ? a.N % 32 == 0 : a.wx_plane > 0)), "bad bf16x6 weight planes");
  if (a.epi == EPI_GEGLU_FWD) {
    RAMP_REQUIRE(!gen && a.N % 256 == 0 && a.aux_out && !a.resid && !a.rowbias,
                 "GEGLU-forward epilogue needs a plain linear with N % 256 == 0 and aux_out");
    RAMP_REQUIRE((x6 && a.wx_packed != 0) == (a.geglu_group == 32), "GEGLU weight tiling: [32 a | 32 g] for the pipelined kernels, [64 a | 64 g] otherwise");
    if (x6) return launch_x6<EPI_GEGLU_FWD, false>(a, s);
    return launch_cfg<2, 2, 2, 2, EPI_GEGLU_FWD, false>(a, 2, s);
  }
  if (a.epi == EPI_GEGLU_BWD) {
    RAMP_REQUIRE(!gen && a.N >= 128 && a.aux_in && !a.resid && !a.rowbias && !a.bias,
                 "GEGLU-backward epilogue needs a plain linear with aux_in");
    if (x6) return launch_x6<EPI_GEGLU_BWD, false>(a, s);
    return launch_cfg<2, 2, 2, 2, EPI_GEGLU_BWD, false>(a, 2, s);
  }
  if (x6) return gen ? launch_x6<EPI_LINEAR, true>(a, s) : launch_x6<EPI_LINEAR, false>(a, s);
  if (a.N >= 128) {                                              // 128 x 128, 2 blocks / CU
    if (!gen) return launch_cfg<2, 2, 2, 2, EPI_LINEAR, false>(a, 2, s);
    return launch_cfg<2, 2, 2, 2, EPI_LINEAR, true>(a, 2, s);
  }
  if (a.N >= 64) return launch_cfg<2, 2, 2, 1, EPI_LINEAR, true>(a, 2, s);      // 128 x 64
  return launch_cfg<4, 1, 1, 1, EPI_LINEAR, true>(a, 3, s);                      // 128 x 32
}

}  // namespace ramp

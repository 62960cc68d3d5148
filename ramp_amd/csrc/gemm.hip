// fp32 MFMA GEMM with conv taps for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains).
//
// Serves every dense contraction on the sampler hot path: the transformer linears
// (layers_attention_mini.py:83-127, 38-45), Conv1d k=5 / k=1 as shifted-tap GEMMs in the
// channels-last layout (layers.py:280-297, 327-361) and all of their dX products
// (UnetInference.py:27 — autograd.grad w.r.t. the input only, weights are frozen).
//
// Tile: BM x BN x 32, 256 threads = 4 waves, each wave owns (MI*32) x (NI*32) as MI*NI
// 32x32 accumulators.  Both operands are K-contiguous ([M][K] activations, [N][K] weights =
// nn.Linear layout), staged global -> registers -> LDS with one 16-byte pad per row, so every
// lane feeds four MFMAs from a single conflict-free ds_read_b128 per operand (the k order
// inside a 32-chunk is permuted identically for A and B).  Double-buffered LDS, one barrier
// per K-tile; the next tile's global loads are in flight during the current tile's MFMAs.
#include "common.h"

namespace ramp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;   // floats per LDS row (16-byte pad: conflict-free ds_read_b128)

template <int WM, int WN, int MI, int NI>
__global__ __launch_bounds__(WM * WN * 64)
void gemm_kernel(GemmArgs a, int tiles_n, int n_tiles, int chunk) {
  constexpr int BM = WM * MI * 32, BN = WN * NI * 32, NT = WM * WN * 64;
  constexpr int AI = BM * 8 / NT, BI = BN * 8 / NT;   // float4 loads per thread per tile
  static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile/thread mismatch");
  extern __shared__ __attribute__((aligned(16))) float smem[];

  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each
  // XCD a contiguous run of logical tiles; N-tiles of one M-tile then hit the same L2.
  const int bid = blockIdx.x;
  const int logical = (bid & 7) * chunk + (bid >> 3);
  if (logical >= n_tiles) return;
  const int tile_n = logical % tiles_n, tile_m = logical / tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int c4 = tid & 7, r0 = tid >> 3;               // staging: float4 column, first row
  constexpr int RSTEP = NT / 8;

  float* As0 = smem;
  float* Bs0 = smem + BM * LDS_LD;
  constexpr int STAGE = (BM + BN) * LDS_LD;

  // per-thread staging rows
  int a_l[AI]; long a_m[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    int m = m0 + r0 + RSTEP * i;
    a_m[i] = (m < a.M) ? m : -1;
    a_l[i] = (m < a.M) ? (m % a.L) : 0;
  }

  const int nk = (a.K + BK - 1) / BK;
  const int total = a.taps * nk;

  f32x4 ra[AI], rb[BI];
  auto load_tile = [&](int it) {
    const int tap = it / nk, k0 = (it - tap * nk) * BK;
    const int sh = a.shift0 + tap * a.shift_step;
    const int kk = k0 + c4 * 4;
    const bool kval = kk < a.K;
    const float* Ab; int ld, kc;
    if (kk < a.K1) { Ab = a.A; ld = a.lda; kc = kk; } else { Ab = a.A2; ld = a.lda2; kc = kk - a.K1; }
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int l = a_l[i] + sh;
      const bool ok = kval && a_m[i] >= 0 && l >= 0 && l < a.L;
      ra[i] = ok ? *reinterpret_cast<const f32x4*>(Ab + (a_m[i] + sh) * ld + kc) : f32x4{0, 0, 0, 0};
    }
    const float* Wb = a.W + (long)tap * a.N * a.K;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int n = n0 + r0 + RSTEP * i;
      rb[i] = (kval && n < a.N) ? *reinterpret_cast<const f32x4*>(Wb + (long)n * a.K + kk) : f32x4{0, 0, 0, 0};
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
      *reinterpret_cast<f32x4*>(Bs + (r0 + RSTEP * i) * LDS_LD + c4 * 4) = rb[i];
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int r = lane & 31, h = lane >> 5;
  const int arow = (wm * MI * 32 + r) * LDS_LD + h * 4;
  const int brow = (wn * NI * 32 + r) * LDS_LD + h * 4;

  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int it = 0; it < total; ++it) {
    const int st = it & 1;
    if (it + 1 < total) load_tile(it + 1);
    const float* As = As0 + st * STAGE;
    const float* Bs = Bs0 + st * STAGE;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 av[MI], bv[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        av[mi] = *reinterpret_cast<const f32x4*>(As + arow + mi * 32 * LDS_LD + q * 8);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        bv[ni] = *reinterpret_cast<const f32x4*>(Bs + brow + ni * 32 * LDS_LD + q * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][e], bv[ni][e], acc[mi][ni], 0, 0, 0);
    }
    if (it + 1 < total) store_tile(st ^ 1);
    __syncthreads();
  }

  // epilogue: lane holds column j = lane&31, rows (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int m = m0 + wm * MI * 32 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (m >= a.M) continue;
      const float* rbp = nullptr;
      if (a.rowbias) rbp = a.rowbias + (long)a.rowvar[a.row0 + m / a.L] * a.rb_stride;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + wn * NI * 32 + ni * 32 + r;
        if (n >= a.N) continue;
        float v = acc[mi][ni][reg];
        if (a.bias) v += a.bias[n];
        if (rbp) v += rbp[n];
        if (a.resid) v += a.resid[(long)m * a.ldr + n];
        if (a.resid2) v += a.resid2[(long)m * a.ldr2 + n];
        if (n < a.N1) a.C[(long)m * a.ldc + n] = v;
        else a.C2[(long)m * a.ldc2 + (n - a.N1)] = v;
      }
    }
  }
}

template <int WM, int WN, int MI, int NI>
static int launch_cfg(const GemmArgs& a, hipStream_t s) {
  constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
  const int n_tiles = tiles_m * tiles_n;
  const int chunk = (n_tiles + 7) / 8;
  const size_t lds = 2 * (size_t)(BM + BN) * LDS_LD * sizeof(float);
  hipLaunchKernelGGL((gemm_kernel<WM, WN, MI, NI>), dim3(chunk * 8), dim3(WM * WN * 64), lds, s, a, tiles_n,
                     n_tiles, chunk);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

template <int WM, int WN, int MI, int NI>
static int set_attr() {
  constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
  const size_t lds = 2 * (size_t)(BM + BN) * LDS_LD * sizeof(float);
  RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<WM, WN, MI, NI>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return 0;
}
int init_gemm_attributes() {
  if (int e = set_attr<2, 2, 2, 2>()) return e;
  if (int e = set_attr<2, 2, 2, 1>()) return e;
  return set_attr<4, 1, 2, 1>();
}

int launch_gemm(const GemmArgs& a_in, hipStream_t s) {
  GemmArgs a = a_in;
  if (a.A2 == nullptr) a.K1 = a.K;
  if (a.C2 == nullptr) a.N1 = a.N;
  if (a.rb_stride == 0) a.rb_stride = a.N;
  RAMP_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.taps >= 1 && a.L >= 1, "bad GEMM dims");
  RAMP_REQUIRE(a.K % 4 == 0 && a.lda % 4 == 0 && a.K1 % 4 == 0, "K, K1 and lda must be multiples of 4 floats");
  RAMP_REQUIRE(a.A2 == nullptr || (a.lda2 % 4 == 0 && a.K1 % BK == 0), "split-K source must start on a K tile");
  RAMP_REQUIRE((reinterpret_cast<uintptr_t>(a.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.W) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(a.A2) & 15) == 0, "A/W must be 16-byte aligned");
  RAMP_REQUIRE(a.rowbias == nullptr || a.rowvar != nullptr, "rowbias needs rowvar");
  if (a.N >= 128) return launch_cfg<2, 2, 2, 2>(a, s);     // 128 x 128
  if (a.N >= 64) return launch_cfg<2, 2, 2, 1>(a, s);      // 128 x 64
  return launch_cfg<4, 1, 2, 1>(a, s);                      // 256 x 32
}

}  // namespace ramp

// Conv1d(k = 5, padding 2) of the narrow residual blocks (C_in, C_out <= 64) as SAMPLE-OWNING waves: the convolutions of
// ResidualTemporalBlock / Conv1dBlock at the two finest levels and the final block (layers.py:280-297, 327-361; UnetInference.py:142-145)
// and their input gradients (the same operator with the taps reversed and the weight transposed).
//
// Why its own kernel.  On the 128-wide tile kernels these layers waste half (N = 64) or three quarters (N = 32, exact-fp32 MFMA) of every
// tile and run at 53-90 TFLOP/s; they carry 10 KFLOP-40 KFLOP per token against 256-512 bytes, i.e. they are HBM-bound work.  Here a wave owns
// T = 48 (or 32) consecutive tokens = whole samples of L tokens (L | T), so the zero padding of the convolution is wave-local: the wave's rows
// go, split into two scaled fp16 planes, into a wave-private LDS tile with two zero rows on either side of every sample; tap j of the
// convolution is the SAME tile read one row further (the B operand of v_mfma_f32_16x16x32_f16: lane (c, g) = channels 32 j' + 8 g .. of token
// c's shifted row).  The whole weight (5 taps x C_out x C_in as fp16 fragment planes, <= 80 KB) is resident in LDS for the block's lifetime --
// no ring, no barrier after the prologue.  D^T[c_out][token] accumulates in 16 x 16 tiles; the next tile's rows are fetched into registers
// while the current tile computes.  fp16x3 products, delayed operand scale / maxima / range guard of ONE call site, like every other GEMM.
#include "args_conv.h"
#include "tokmma.h"

#include <algorithm>

namespace ramp {

namespace {

// rows of a wave's tile: T tokens + 2 zero rows in front of every sample and behind the last.  T = 48 / 32 (NG 3 / 2): L >= 8, <= 6 samples;
// T = 64 (NG 4, round 6: the finest level of the H = 64 configurations): ONE sample of L = 64 tokens
constexpr int tc_xrows(int NG) { return NG == 4 ? 64 + 2 + 2 : 48 + 2 * 6 + 2; }
// bytes of wave-private scratch behind the tiles (GroupNorm sums: 8 groups x T tokens, then 8 x samples results)
constexpr int tc_scr(int NG) { return NG == 4 ? 2304 : 2048; }

__device__ __forceinline__ float tc_mish(float x) {     // rowops.hip mish_f
  const float e = __builtin_amdgcn_exp2f(fminf(x, 20.f) * 1.44269504088896340736f);
  const float n = e * (e + 2.f);
  return x * (n * __builtin_amdgcn_rcpf(n + 2.f));
}
__device__ __forceinline__ float tc_mish_grad(float x) {   // rowops.hip mish_grad_f
  const float e = __builtin_amdgcn_exp2f(fminf(x, 20.f) * 1.44269504088896340736f);
  const float n = e * (e + 2.f);
  const float w = __builtin_amdgcn_rcpf(n + 2.f);
  return n * w + x * (4.f * e * (e + 1.f) * w * w);
}

__device__ __forceinline__ f32x4 tc_mm32(const u32x4 a, const u32x4 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void tc_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

}  // namespace

// W [5][N][K] fp32 (tap, c_out, c_in) -> out [tap][N / 16][K / 32][plane 2][lane 64][8 halves]: the A fragment of v_mfma_f32_16x16x32_f16
// (lane (r, kq): row 16 nb + r, k = 32 j + 8 kq .. + 7)
__global__ void tkc_pack_kernel(const float* __restrict__ W, unsigned short* __restrict__ out, int N, int K, float scale) {
  const int total = 5 * (N / 16) * (K / 32) * 64;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63, f = idx >> 6;
  const int j = f % (K / 32), nb = (f / (K / 32)) % (N / 16), tap = f / ((K / 32) * (N / 16));
  const float* src = W + ((long)tap * N + 16 * nb + (lane & 15)) * K + 32 * j + 8 * (lane >> 4);
  unsigned short* o = out + (long)f * 1024 + lane * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = src[e] * scale;
    const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
    o[e] = __builtin_bit_cast(unsigned short, hi); o[512 + e] = __builtin_bit_cast(unsigned short, lo);
  }
}
int tkc_pack(const float* W, int N, int K, float scale, unsigned short* out, hipStream_t s) {
  RAMP_REQUIRE(W && out && N % 16 == 0 && K % 32 == 0 && N >= 16 && N <= 64 && K >= 32 && K <= 64, "tkc_pack: bad shape");
  const int total = 5 * (N / 16) * (K / 32) * 64;
  hipLaunchKernelGGL(tkc_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, s, W, out, N, K, scale);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}
size_t tkc_packed_halves(int N, int K) { return (size_t)5 * (N / 16) * (K / 32) * 1024; }

// NG: 16-token groups per wave; NB = N / 16 output blocks; KS = K / 32 k32 steps.  Round 5: the GroupNorm(8) + Mish around the convolution fused like in
// tkw.hip -- EPI: Cst = conv + bias (the stash), its statistics, Y = mish(GN(Cst)) + time bias + residual; PRO: the operand is the GroupNorm + Mish input
// gradient GNbwd(X (.) mish'(gamma x^ + beta) gamma) of the forward layer's stash gn_c.  A wave owns whole samples and ALL channels, so both reductions
// (over a sample's tokens x a group's C / 8 channels) stay inside the wave: in-lane over the four channels of a register quad, one shuffle for the second
// quad of an 8-channel group, the tokens through 2 KB of wave-private scratch.
template <int NG, int NB, int KS, int PRO, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
void tkc_kernel(TkcArgs a, int n_tiles) {
  constexpr int T = 16 * NG, K = 32 * KS, N = 16 * NB;
  constexpr int XROW = 4 * K + 16;                          // bytes of a tile row: hi plane (2 K) | lo plane (2 K) | 16 bytes of padding (bank spread)
  constexpr int WBYTES = 5 * NB * KS * 2048;
  constexpr int TC_XROWS = tc_xrows(NG), TC_SCR = tc_scr(NG);
  static_assert(8 * T * 4 + 8 * (NG == 4 ? 1 : 6) * 4 <= TC_SCR, "GroupNorm scratch");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int n_my = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

  // amax_in == nullptr (a job's calibration evaluation, ramp_op_* without a previous maximum): no delayed scale exists, so every
  // wave tile is scaled EXACTLY from its own maximum (max -> [2^5, 2^6), as the attention-internal operands of atk.hip): range-free,
  // whatever the operand's magnitude -- an unscaled split would lose the low plane below ~1e-4 and overflow above 65504
  const bool exact = a.amax_in == nullptr;
  float s_in = scale_of(a.amax_in);
  float os = a.wsi / s_in;
  float amax = 0.f;

  // ---- prologue: the weight planes -> LDS (once), the wave's tile zeroed (its padding rows stay zero for the block's lifetime)
  for (int i = tid; i < WBYTES / 16; i += 256) reinterpret_cast<u32x4*>(smem)[i] = reinterpret_cast<const u32x4*>(a.W)[i];
  char* xt = smem + WBYTES + wave * (TC_XROWS * XROW);
  for (int i = lane; i < TC_XROWS * XROW / 16; i += 64) reinterpret_cast<u32x4*>(xt)[i] = u32x4{0u, 0u, 0u, 0u};
  float* const wscr = reinterpret_cast<float*>(smem + WBYTES + 4 * (TC_XROWS * XROW) + wave * TC_SCR);
  __syncthreads();

  // rows of the wave's tokens in its tile: token t of sample s = t / L sits in row t + 2 s + 2
  int xrow[NG], smp[NG];
#pragma unroll
  for (int t = 0; t < NG; ++t) { smp[t] = (16 * t + c) / a.L; xrow[t] = (16 * t + c) + 2 * smp[t] + 2; }
  const int n_smp = T / a.L;                                // samples of a wave tile
  // GroupNorm bookkeeping (PRO: over the K operand channels, EPI: over the N output channels; 8 groups either way): a channel block of 16 holds
  // 16 / (C / 8) groups -- C = 32: four (one per register quad g), C = 64: two (quads 2 j, 2 j + 1)
  auto tok_sums = [&](float (&v)[4][NG], int nblk, int C) __attribute__((always_inline)) {
    // v[b][t]: this lane's sum over its four channels of block b (b < nblk), token 16 t + c  ->  per-(sample, group) mean of the summed quantity
    const int gsz = C >> 3;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (b < nblk) {
#pragma unroll
        for (int t = 0; t < NG; ++t) {
          float q = v[b][t];
          if (gsz == 8) q += __shfl_xor(q, 16);             // the group's other register quad
          const int grp = gsz == 4 ? 4 * b + g : 2 * b + (g >> 1);
          if (gsz == 4 || !(g & 1)) wscr[grp * T + 16 * t + c] = q;
        }
      }
    }
    tc_wave_sync();
    if (lane < 8 * n_smp) {                                 // one lane per (group, sample): the sample's L tokens
      const int grp = lane / n_smp, sm = lane - grp * n_smp;
      float q = 0.f;
      for (int u = 0; u < a.L; ++u) q += wscr[grp * T + sm * a.L + u];
      wscr[8 * T + lane] = q / (float)(a.L * gsz);
    }
    tc_wave_sync();
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (b < nblk) {
#pragma unroll
        for (int t = 0; t < NG; ++t) {
          const int grp = gsz == 4 ? 4 * b + g : 2 * b + (g >> 1);
          v[b][t] = wscr[8 * T + grp * n_smp + smp[t]];
        }
      }
    }
    tc_wave_sync();
  };
  const int m_last = a.M - 1;
  const char* xbase = reinterpret_cast<const char*>(a.X);

  // raw rows of a tile: lane (c, g) holds channels 16 fb + 4 g .. + 3 of token 16 t + c
  f32x4 xr[K / 16][NG];
  f32x4 cr[PRO ? K / 16 : 1][PRO ? NG : 1];                 // PRO: the same elements of the forward layer's stash, then x^
  float gmu[PRO ? K / 16 : 1][PRO ? NG : 1], grs[PRO ? K / 16 : 1][PRO ? NG : 1];      // PRO: mean / rstd of this lane's (sample, group)s
  auto load_x = [&](int tile) __attribute__((always_inline)) {
    int tk0 = tile * (4 * T) + wave * T + c;
    asm volatile("" : "+v"(tk0));
#pragma unroll
    for (int t = 0; t < NG; ++t) {
      const unsigned tk = (unsigned)min(tk0 + 16 * t, m_last);
      const char* row = xbase + ((size_t)tk * (unsigned)(4 * a.ldx) + 16u * (unsigned)g);
#pragma unroll
      for (int fb = 0; fb < K / 16; ++fb) xr[fb][t] = *reinterpret_cast<const f32x4*>(row + 64 * fb);
      if constexpr (PRO) {
        const char* crow = reinterpret_cast<const char*>(a.gn_c) + ((size_t)tk * (unsigned)(4 * K) + 16u * (unsigned)g);
        const unsigned srow = tk / (unsigned)a.L;
#pragma unroll
        for (int fb = 0; fb < K / 16; ++fb) {
          cr[fb][t] = *reinterpret_cast<const f32x4*>(crow + 64 * fb);
          const int grp = K == 32 ? 4 * fb + g : 2 * fb + (g >> 1);
          gmu[fb][t] = a.gn_stats[((size_t)srow * 8 + grp) * 2]; grs[fb][t] = a.gn_stats[((size_t)srow * 8 + grp) * 2 + 1];
        }
      }
    }
  };
  f32x4 pgam[PRO ? K / 16 : 1], pbet[PRO ? K / 16 : 1];
  if constexpr (PRO) {
#pragma unroll
    for (int fb = 0; fb < K / 16; ++fb) {
      pgam[fb] = *reinterpret_cast<const f32x4*>(a.gn_gamma + 16 * fb + 4 * g); pbet[fb] = *reinterpret_cast<const f32x4*>(a.gn_beta + 16 * fb + 4 * g);
    }
  }
  load_x((int)blockIdx.x);

#pragma unroll 1
  for (int ti = 0; ti < n_my; ++ti) {
    const int tile = (int)blockIdx.x + ti * (int)gridDim.x;
    const int tile_n = ti + 1 < n_my ? tile + (int)gridDim.x : tile;
    const int tok0 = tile * (4 * T) + wave * T;
    const bool full = tok0 + T <= a.M;

    if constexpr (PRO) {
      // the operand becomes the GroupNorm + Mish input gradient: d = dy mish'(gamma x^ + beta) gamma; dc = (d - mean(d) - x^ mean(d x^)) rstd
      float v1[4][NG], v2[4][NG];
#pragma unroll
      for (int fb = 0; fb < K / 16; ++fb)
#pragma unroll
        for (int t = 0; t < NG; ++t) {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float h = (cr[fb][t][j] - gmu[fb][t]) * grs[fb][t];
            const float d = xr[fb][t][j] * tc_mish_grad(h * pgam[fb][j] + pbet[fb][j]) * pgam[fb][j];
            cr[fb][t][j] = h; xr[fb][t][j] = d;
            s1 += d; s2 += d * h;
          }
          v1[fb][t] = s1; v2[fb][t] = s2;
        }
      tok_sums(v1, K / 16, K);
      tok_sums(v2, K / 16, K);
#pragma unroll
      for (int fb = 0; fb < K / 16; ++fb)
#pragma unroll
        for (int t = 0; t < NG; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) xr[fb][t][j] = (xr[fb][t][j] - v1[fb][t] - cr[fb][t][j] * v2[fb][t]) * grs[fb][t];
    }
    if (exact) {
      float tm = 0.f;
#pragma unroll
      for (int t = 0; t < NG; ++t) {
        const float lm = (full || tok0 + 16 * t + c < a.M) ? 1.f : 0.f;
#pragma unroll
        for (int fb = 0; fb < K / 16; ++fb) tm = amax4(xr[fb][t] * lm, tm);
      }
      tm = fmaxf(tm, __shfl_xor(tm, 32)); tm = fmaxf(tm, __shfl_xor(tm, 16)); tm = fmaxf(tm, __shfl_xor(tm, 8));
      tm = fmaxf(tm, __shfl_xor(tm, 4)); tm = fmaxf(tm, __shfl_xor(tm, 2)); tm = fmaxf(tm, __shfl_xor(tm, 1));
      s_in = scale_from(tm);
      os = a.wsi / s_in;
    }
    // ---- rows -> recorded maximum, two scaled fp16 planes -> the wave's LDS tile
#pragma unroll
    for (int t = 0; t < NG; ++t) {
      const float lm = (full || tok0 + 16 * t + c < a.M) ? 1.f : 0.f;      // tokens past M: zero rows (they are neighbours of nobody: whole samples)
      char* dst = xt + xrow[t] * XROW + 8 * g;
#pragma unroll
      for (int fb = 0; fb < K / 16; ++fb) {
        const f32x4 v = xr[fb][t] * lm;
        amax = amax4(v, amax);
        unsigned h0, h1, l0, l1;
        split4(v * s_in, h0, h1, l0, l1);
        *reinterpret_cast<u32x2*>(dst + 32 * fb) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(dst + 2 * K + 32 * fb) = u32x2{l0, l1};
      }
    }
    tc_wave_sync();
    load_x(tile_n);                                         // the next tile's rows: in flight during this tile's products

    // ---- D^T[c_out][token] = sum_tap W_tap X_shifted^T: tap j reads the tile dir * (j - 2) rows further
    f32x4 acc[NB][NG];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int t = 0; t < NG; ++t) acc[nb][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 5; ++tap) {
      const int sh = a.dir * (tap - 2);
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        u32x4 bh[NG], bl[NG];
#pragma unroll
        for (int t = 0; t < NG; ++t) {
          const char* p = xt + (xrow[t] + sh) * XROW + 64 * j + 16 * g;
          bh[t] = *reinterpret_cast<const u32x4*>(p);
          bl[t] = *reinterpret_cast<const u32x4*>(p + 2 * K);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const char* wp = smem + ((tap * NB + nb) * KS + j) * 2048 + lane * 16;
          const u32x4 wh = *reinterpret_cast<const u32x4*>(wp), wl = *reinterpret_cast<const u32x4*>(wp + 1024);
#pragma unroll
          for (int t = 0; t < NG; ++t) {
            f32x4 v = acc[nb][t];
            v = tc_mm32(wh, bl[t], v);
            v = tc_mm32(wl, bh[t], v);
            v = tc_mm32(wh, bh[t], v);
            acc[nb][t] = v;
          }
        }
      }
    }
    tc_wave_sync();                                         // (the tile is rewritten at the top of the next iteration)

    if constexpr (EPI) {
      // ---- epilogue with GroupNorm(8) + Mish: c = acc os + bias (the stash), mean / variance per (sample, group) in two passes like gn_fwd_kernel,
      // y = mish((c - mean) rstd gamma + beta) + time bias + residual
      unsigned trow[NG];
#pragma unroll
      for (int t = 0; t < NG; ++t) trow[t] = (unsigned)min(tok0 + 16 * t + c, m_last);
      float mean[4][NG], var[4][NG];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const f32x4 bq = *reinterpret_cast<const f32x4*>(a.bias + 16 * nb + 4 * g);
#pragma unroll
        for (int t = 0; t < NG; ++t) {
          acc[nb][t] = acc[nb][t] * os + bq;
          if (full || tok0 + 16 * t + c < a.M) *reinterpret_cast<f32x4*>(a.Cst + (size_t)trow[t] * N + 16 * nb + 4 * g) = acc[nb][t];
          mean[nb][t] = (acc[nb][t][0] + acc[nb][t][1]) + (acc[nb][t][2] + acc[nb][t][3]);
        }
      }
      tok_sums(mean, NB, N);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int t = 0; t < NG; ++t) {
          float q = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float d = acc[nb][t][j] - mean[nb][t]; q += d * d; }
          var[nb][t] = q;
        }
      tok_sums(var, NB, N);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const f32x4 gam = *reinterpret_cast<const f32x4*>(a.gamma + 16 * nb + 4 * g), bet = *reinterpret_cast<const f32x4*>(a.beta + 16 * nb + 4 * g);
        const f32x4 tb = a.tbias ? *reinterpret_cast<const f32x4*>(a.tbias + 16 * nb + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NG; ++t) {
          const float rstd = 1.f / sqrtf(var[nb][t] + a.eps);
          const bool live = full || tok0 + 16 * t + c < a.M;
          // one writer per (sample, group): the sample's first token, the group's first register quad
          if (live && (16 * t + c) - smp[t] * a.L == 0 && (N == 32 || !(g & 1))) {
            const int grp = N == 32 ? 4 * nb + g : 2 * nb + (g >> 1);
            float* st = a.stats + ((size_t)(trow[t] / (unsigned)a.L) * 8 + grp) * 2;
            st[0] = mean[nb][t]; st[1] = rstd;
          }
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = tc_mish((acc[nb][t][j] - mean[nb][t]) * rstd * gam[j] + bet[j]) + tb[j];
          if (a.resid) v += *reinterpret_cast<const f32x4*>(a.resid + (size_t)trow[t] * a.ldr + 16 * nb + 4 * g);
          if (live) *reinterpret_cast<f32x4*>(a.Y + (size_t)trow[t] * a.ldy + 16 * nb + 4 * g) = v;
        }
      }
    } else {
    // ---- epilogue: y[tok][16 nb + 4 g ..] = acc * os + bias + resid + resid2.  Full wave tiles: every load first, every store unconditional
    // (a store behind a per-lane predicate sits in its own basic block behind s_waitcnt vmcnt(0): one memory round trip per store)
    {
      unsigned trow[NG];
#pragma unroll
      for (int t = 0; t < NG; ++t) trow[t] = (unsigned)min(tok0 + 16 * t + c, m_last);
      f32x4 bq[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bq[nb] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + 16 * nb + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
      if (full) {
        f32x4 rz[NB][NG];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int t = 0; t < NG; ++t) rz[nb][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.resid) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int t = 0; t < NG; ++t) rz[nb][t] = *reinterpret_cast<const f32x4*>(a.resid + (size_t)trow[t] * a.ldr + 16 * nb + 4 * g);
        }
        if (a.resid2) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int t = 0; t < NG; ++t) rz[nb][t] += *reinterpret_cast<const f32x4*>(a.resid2 + (size_t)trow[t] * a.ldr2 + 16 * nb + 4 * g);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int t = 0; t < NG; ++t) *reinterpret_cast<f32x4*>(a.Y + (size_t)trow[t] * a.ldy + 16 * nb + 4 * g) = acc[nb][t] * os + bq[nb] + rz[nb][t];
      } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int t = 0; t < NG; ++t) {
            f32x4 v = acc[nb][t] * os + bq[nb];
            if (a.resid) v += *reinterpret_cast<const f32x4*>(a.resid + (size_t)trow[t] * a.ldr + 16 * nb + 4 * g);
            if (a.resid2) v += *reinterpret_cast<const f32x4*>(a.resid2 + (size_t)trow[t] * a.ldr2 + 16 * nb + 4 * g);
            if (tok0 + 16 * t + c < a.M) *reinterpret_cast<f32x4*>(a.Y + (size_t)trow[t] * a.ldy + 16 * nb + 4 * g) = v;
          }
      }
    }
  }

    }
  amax = fmaxf(amax, __shfl_xor(amax, 32)); amax = fmaxf(amax, __shfl_xor(amax, 16)); amax = fmaxf(amax, __shfl_xor(amax, 8));
  amax = fmaxf(amax, __shfl_xor(amax, 4)); amax = fmaxf(amax, __shfl_xor(amax, 2)); amax = fmaxf(amax, __shfl_xor(amax, 1));
  // ONE atomic per block, behind a plain read of the slot (2048 same-address atomics at the tail of a 20-40 us launch cost 20 us: core.h)
  __syncthreads();
  // (exact mode: every tile was scaled from its own maximum -- nothing can be out of range, and s_in is just the LAST tile's scale: no guard)
  record_amax_block_guarded<true>(a.amax_out, amax, reinterpret_cast<float*>(smem + WBYTES), exact ? nullptr : a.range_flag, s_in, a.site);
}

bool tkc_applicable(int M, int L, int N, int K, int* ng) {
  int n = 0;
  if (L >= 8 && 48 % L == 0) n = 3; else if (L >= 8 && 32 % L == 0) n = 2; else if (L == 64 && !(N == 64 && K == 64)) n = 4;      // (64 x 64 at L = 64: 164 KB of LDS; no such layer)
  if (ng) *ng = n;
  return n != 0 && M > 0 && M % L == 0 && (N == 32 || N == 64) && (K == 32 || K == 64);
}

namespace {
template <int NG, int NB, int KS> size_t tkc_lds() { return (size_t)5 * NB * KS * 2048 + 4 * (size_t)tc_xrows(NG) * (4 * 32 * KS + 16) + 4 * tc_scr(NG); }
template <int NG, int NB, int KS, int PRO, int EPI> int tkc_go(const TkcArgs& a, int n_tiles, hipStream_t s) {
  const size_t lds = tkc_lds<NG, NB, KS>();
  const int per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;         // blocks the LDS lets a CU hold (a block pays an 20-80 KB weight prologue: no more blocks than stay resident)
  const int nb = std::min(n_tiles, per_cu * device_cu_count());
  hipLaunchKernelGGL((tkc_kernel<NG, NB, KS, PRO, EPI>), dim3(nb), dim3(256), lds, s, a, n_tiles);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}
}  // namespace

int launch_tkc(const TkcArgs& a, hipStream_t s) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  int ng = 0;
  RAMP_REQUIRE(tkc_applicable(a.M, a.L, a.N, a.K, &ng), "tkc: C_in, C_out in {32, 64}, tokens per sample >= 8 dividing 48 or 32 (or 64, not 64 x 64 channels), whole samples");
  RAMP_REQUIRE(a.X && a.W && a.Y && (a.dir == 1 || a.dir == -1), "tkc: bad operand");
  RAMP_REQUIRE(al16(a.X) && al16(a.W) && al16(a.Y) && al16(a.bias) && al16(a.resid) && al16(a.resid2) && a.ldx % 4 == 0 && a.ldy % 4 == 0 &&
               a.ldr % 4 == 0 && a.ldr2 % 4 == 0 && a.ldx >= a.K && a.ldy >= a.N, "tkc: operands must be 16-byte aligned");
  RAMP_REQUIRE((long)a.M * a.ldx * 4 < (1l << 32), "tkc: 32-bit row offsets");
  RAMP_REQUIRE(!ranges_overlap(a.Y, ((size_t)(a.M - 1) * a.ldy + a.N) * 4, a.X, ((size_t)(a.M - 1) * a.ldx + a.K) * 4), "tkc: the output must not overlap the operand");
  const int T = 16 * ng, n_tiles = (a.M + 4 * T - 1) / (4 * T);
  const int pro = a.gn_c ? 1 : 0, epi = a.Cst ? 1 : 0;
  RAMP_REQUIRE(!(pro && epi), "tkc: the GroupNorm backward operand and the GroupNorm epilogue do not occur together");
  if (pro) RAMP_REQUIRE(a.gn_stats && a.gn_gamma && a.gn_beta && al16(a.gn_c) && al16(a.gn_gamma) && al16(a.gn_beta) && a.ldx >= a.K, "tkc: GroupNorm-backward operand incomplete");
  if (epi) RAMP_REQUIRE(a.stats && a.gamma && a.beta && a.bias && !a.resid2 && al16(a.Cst) && al16(a.gamma) && al16(a.beta) && al16(a.tbias), "tkc: GroupNorm epilogue incomplete");
#define TC_SHAPE(NGV, PROV, EPIV) \
    if (a.N == 64 && a.K == 64) return tkc_go<NGV, 4, 2, PROV, EPIV>(a, n_tiles, s); \
    if (a.N == 64 && a.K == 32) return tkc_go<NGV, 4, 1, PROV, EPIV>(a, n_tiles, s); \
    if (a.N == 32 && a.K == 64) return tkc_go<NGV, 2, 2, PROV, EPIV>(a, n_tiles, s); \
    return tkc_go<NGV, 2, 1, PROV, EPIV>(a, n_tiles, s);
#define TC_CASE(NGV) \
  if (ng == NGV) { \
    if (pro) { TC_SHAPE(NGV, 1, 0) } \
    if (epi) { TC_SHAPE(NGV, 0, 1) } \
    TC_SHAPE(NGV, 0, 0) \
  }
  TC_CASE(3) TC_CASE(2)
#undef TC_SHAPE
#define TC_SHAPE(NGV, PROV, EPIV) \
    if (a.N == 64 && a.K == 32) return tkc_go<NGV, 4, 1, PROV, EPIV>(a, n_tiles, s); \
    if (a.N == 32 && a.K == 64) return tkc_go<NGV, 2, 2, PROV, EPIV>(a, n_tiles, s); \
    if (a.N == 32 && a.K == 32) return tkc_go<NGV, 2, 1, PROV, EPIV>(a, n_tiles, s);
  TC_CASE(4)
#undef TC_CASE
#undef TC_SHAPE
  RAMP_REQUIRE(false, "tkc: variant not built");
}

int init_tkc_attributes() {
#define TC_ATTR1(NGV, NBV, KSV, PROV, EPIV) RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tkc_kernel<NGV, NBV, KSV, PROV, EPIV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tkc_lds<NGV, NBV, KSV>()))
#define TC_ATTR(NGV, NBV, KSV) TC_ATTR1(NGV, NBV, KSV, 0, 0); TC_ATTR1(NGV, NBV, KSV, 1, 0); TC_ATTR1(NGV, NBV, KSV, 0, 1)
  TC_ATTR(3, 4, 2); TC_ATTR(3, 4, 1); TC_ATTR(3, 2, 2); TC_ATTR(3, 2, 1);
  TC_ATTR(2, 4, 2); TC_ATTR(2, 4, 1); TC_ATTR(2, 2, 2); TC_ATTR(2, 2, 1);
  TC_ATTR(4, 4, 1); TC_ATTR(4, 2, 2); TC_ATTR(4, 2, 1);
#undef TC_ATTR
#undef TC_ATTR1
  return 0;
}

}  // namespace ramp

// Conv1d(k = 5, padding 2) of the WIDE residual blocks (C_out a multiple of 128) with GroupNorm(8) + Mish fused around it, as SAMPLE-OWNING
// BLOCKS: the convolutions of ResidualTemporalBlock / Conv1dBlock at the two coarsest levels, the middle blocks and the first up level
// (layers.py:280-297, 327-361) and their input gradients.
//
//   forward  (EPI 1):  C = conv5(X) + bias  (kept: the VJP stash);  stats = GroupNorm(8) mean / rstd of C per (sample, group);
//                      Y = Mish(GN(C) gamma + beta) + time bias + residual                       -- one launch instead of conv + gn_fwd
//   backward (PRO 1):  Y = conv5^T( GNbwd( dY (.) Mish'(gamma x^ + beta) gamma ; x^ ) ) + resid + resid2     -- one launch instead of gn_bwd + conv
//
// Why its own kernel.  On the 128 x 128 tile kernels these layers re-stage (global load -> scale -> split -> LDS) the activation tile once per
// TAP and per 128 / 256 output columns, run at 0.22 of the fp16x3 ceiling, and their GroupNorms are a kernel pair of their own (11 GB of HBM
// traffic per evaluation).  Here a 4-wave block owns 96 consecutive tokens = WHOLE samples of L tokens (L | 96):
//   * the block's rows go ONCE, as two scaled fp16 planes, into an LDS tile with two zero rows between samples (row(t) = t + 2 (t / L) + 2):
//     tap j of the convolution is the SAME tile read j - 2 rows further, the zero padding of the convolution is the padding of the tile;
//   * the waves split the OUTPUT CHANNELS (wave w: 32 MB channels of every 128 MB): D^T[c_out][token] = sum_tap W_tap X_shifted^T on
//     v_mfma_f32_32x32x16_f16, the weight fragment = A operand straight from global memory / L2 in the tile kernels' packed planes
//     (launch_pack_h3: no new weight layout), the 96 tokens = B operand from LDS; no barrier between the staging and the epilogue;
//   * GroupNorm statistics are per (sample, group of C / 8 channels): a group's channels lie inside ONE wave's slice and the block holds the
//     whole sample, so mean / variance (two passes over the accumulators, like gn_fwd_kernel) never leave the wave;
//   * the GroupNorm BACKWARD is folded into the operand staging of the input-gradient convolution: a wave stages whole samples
//     (24 tokens = 24 / L samples), the two sums of the backward formula are in-wave shuffles.
// Operand channels beyond 256 are staged in chunks (K = 512: the concatenated input of ups.0.0, two sources); outputs beyond 128 MB channels
// run as passes over the same staged tile (N = 512: the split input gradient of ups.0.0).
// fp16x3 products, delayed operand scale / recorded maximum / range guard of ONE call site, like every other GEMM of the library.
#include "args_conv.h"
#include "tokmma.h"

#include <algorithm>

namespace ramp {

namespace {

constexpr int TW_TB = 96;                                   // tokens of a block tile
constexpr int TW_NT = 3;                                    // = 32-token groups (the B operand's columns)
constexpr int TW_SCRATCH = 4096 + 4 * 1024;                 // behind the tile: 1 KB per wave + bias / gamma / beta / time bias of <= 256 channels

__device__ __forceinline__ float tw_mish(float x) {         // rowops.hip mish_f: x n / (n + 2), n = e (e + 2), e = exp(x)
  const float e = __builtin_amdgcn_exp2f(fminf(x, 20.f) * 1.44269504088896340736f);
  const float n = e * (e + 2.f);
  return x * (n * __builtin_amdgcn_rcpf(n + 2.f));
}
__device__ __forceinline__ float tw_mish_grad(float x) {    // rowops.hip mish_grad_f
  const float e = __builtin_amdgcn_exp2f(fminf(x, 20.f) * 1.44269504088896340736f);
  const float n = e * (e + 2.f);
  const float w = __builtin_amdgcn_rcpf(n + 2.f);
  return n * w + x * (4.f * e * (e + 1.f) * w * w);
}
__device__ __forceinline__ void tw_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

struct TwGeom { int KC, n_chunks, rows, xrow, n_pass; size_t lds; };
// the GroupNorm-backward scratch holds 8 groups x 8 instructions: groups of a chunk = (KC / 4) / (K / 32) <= 8
#define LR_GROUPS_OK(KC_, K_) (((KC_) / 4) / ((K_) / 32) <= 8)

}  // namespace

// MB: 32-channel blocks of a wave per pass (1: N = 128, 2: N = 256 / 512); PRO > 0: GroupNorm backward folded into the operand, PRO = the wave
// instructions of RP rows of one staging pass (3, 4, 6, 8: whole samples); EPI 1: GroupNorm + Mish behind the convolution
template <int MB, int PRO, int EPI>
__device__ __forceinline__ void tkw_body(const TkwArgs& a, int n_tiles, int kc32, int n_chunks, int n_rows, int n_pass, int mulL) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tok = lane & 31, kh = lane >> 5;
  const int KC = kc32 << 5;                                 // (a multiple of 32: the plane offset 2 KC keeps 16-byte LDS reads aligned)
  const int XROW = 4 * KC + 16;                             // hi plane (2 KC bytes) | lo plane (2 KC) | 16 bytes (bank spread; 32 .. 112 measured the same)
  const int KS = KC >> 4;                                   // k16 steps of a chunk
  char* const scr = smem + (size_t)n_rows * XROW;           // small scratch behind the tile: 4 waves x 1 KB
  float* const wscr = reinterpret_cast<float*>(scr + wave * 1024);

  const float s_in = scale_of(a.amax_in);
  const float os = a.wsi / s_in;
  float amax = 0.f;
  // t / L for 0 <= t < 96 without a division: mulL = ceil(65536 / L) (exact for these t: the error t (mulL - 65536 / L) / 65536 < 1 / L)
  auto divL = [&](int t) __attribute__((always_inline)) { return (t * mulL) >> 16; };
  // row of token t in the tile: two zero rows in front of every sample (and behind the last)
  auto trow = [&](int t) __attribute__((always_inline)) { return t + 2 * divL(t) + 2; };

  // the tile's padding rows are zeroed once and never written again
  for (int i = tid; i < n_rows * XROW / 16; i += 256) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0u, 0u, 0u, 0u};
  float* const prm = reinterpret_cast<float*>(scr + 4096);  // EPI 1: [bias | gamma | beta | time bias] of the N <= 256 channels
  if (EPI == 1) {
    for (int i = tid; i < a.N; i += 256) {
      prm[i] = a.bias[i]; prm[256 + i] = a.gamma[i]; prm[512 + i] = a.beta[i]; prm[768 + i] = a.tbias ? a.tbias[i] : 0.f;
    }
  }
  __syncthreads();

  // B-operand rows of this lane's tokens: token t = 32 tg + tok sits in row t + 2 (t / L) + 2
  int boff[TW_NT];
#pragma unroll
  for (int tg = 0; tg < TW_NT; ++tg) boff[tg] = trow(32 * tg + tok) * XROW + 16 * kh;
  const int LR = KC >> 2;                                   // staging: lanes per token row (a float4 each), rows per wave instruction
  const int RP = 64 / LR;
  const int sl = lane % LR, sr = lane / LR;
  const int m_last = a.M - 1;

#pragma unroll 1
  for (int tile = (int)blockIdx.x; tile < n_tiles; tile += (int)gridDim.x) {
    const int tok0 = tile * TW_TB;
#pragma unroll 1
    for (int pass = 0; pass < n_pass; ++pass) {
      f32x16 acc[MB][TW_NT];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int tg = 0; tg < TW_NT; ++tg)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[mb][tg][i] = 0.f;
      const int nb0 = pass * 128 * MB + wave * 32 * MB;     // this wave's first output channel of the pass

#pragma unroll 1
      for (int kc = 0; kc < n_chunks; ++kc) {
        if (pass > 0 || kc > 0 || tile != (int)blockIdx.x) __syncthreads();      // every wave is done reading the tile
        if (n_chunks > 1 || pass == 0) {
          // ---- staging: the wave's 24 tokens x KC channels -> (optionally GroupNorm-backward) -> maximum, two scaled fp16 planes -> tile
          const int ch = kc * KC + 4 * sl;                  // this lane's four channels
          if constexpr (PRO == 0) {
#pragma unroll 1
            for (int it0 = 0; it0 < 24; it0 += 8 * RP) {
              f32x4 v[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                const int t = 24 * wave + it0 + u * RP + sr;
                const int tk = min(tok0 + t, m_last);
                const float* src = ch < a.K1 ? a.X + (size_t)tk * a.ldx + ch : a.X2 + (size_t)tk * a.ldx2 + (ch - a.K1);
                v[u] = (it0 + u * RP < 24 && !(a.ablate & 2)) ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
              }
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                if (it0 + u * RP < 24) {
                  const int t = 24 * wave + it0 + u * RP + sr;
                  const f32x4 x = (tok0 + t < a.M) ? v[u] : f32x4{0.f, 0.f, 0.f, 0.f};
                  amax = amax4(x, amax);
                  unsigned h0, h1, l0, l1;
                  split4(x * s_in, h0, h1, l0, l1);
                  char* dst = smem + trow(t) * XROW + 8 * sl;
                  *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
                  *reinterpret_cast<u32x2*>(dst + 2 * KC) = u32x2{l0, l1};
                }
              }
            }
          } else {
            // GroupNorm backward on whole samples: d = dy mish'(gamma x^ + beta) gamma; dc = (d - mean(d) - x^ mean(d x^)) rstd, the two
            // means over the sample's L tokens x (K / 8) channels of the group.  The wave's samples (dealt: wave w takes [w n / 4, (w + 1) n / 4)
            // of the tile's n) are staged in passes of PRO wave instructions of RP rows = whole samples; a pass's rows are requested at once
            // (one memory round trip per pass, not per sample).  Each instruction's partial sums are reduced over the group's lanes and
            // the row slots by shuffles and meet in the wave's scratch, where one lane per (group, sample) adds the sample's instructions
            constexpr int NW = PRO > 0 ? PRO : 1;           // wave instructions per pass held in registers (a multiple of those per sample)
            const int G4 = (a.K >> 5);                      // lanes of a group: (K / 8) / 4
            const int grp = ch / (a.K >> 3), gl = sl / G4;  // the group among the 8, among this chunk's
            const int n_gl = LR / G4;
            const f32x4 gam = *reinterpret_cast<const f32x4*>(a.gn_gamma + ch), bet = *reinterpret_cast<const f32x4*>(a.gn_beta + ch);
            const float inv_cnt = 1.f / (float)(a.L * (a.K >> 3));
            const int n_it = a.L / RP;                      // wave instructions per sample (RP | L is checked on the host)
            const int spp = NW / n_it;                      // samples per pass
            const int n_smp = TW_TB / a.L;
            const int sm_lo = (wave * n_smp) >> 2, sm_hi = ((wave + 1) * n_smp) >> 2;
#pragma unroll 1
            for (int sm0 = sm_lo; sm0 < sm_hi; sm0 += spp) {
              const int t_lo = sm0 * a.L;
              f32x4 dv[NW], hv[NW];
              float mu[NW], rs[NW];
              int u_s = 0, u_i = 0;                          // sample of instruction u inside the pass, position inside the sample
#pragma unroll
              for (int u = 0; u < NW; ++u) {
                const int tk = min(tok0 + t_lo + u * RP + sr, m_last);
                dv[u] = *reinterpret_cast<const f32x4*>(a.X + (size_t)tk * a.ldx + ch);
                hv[u] = *reinterpret_cast<const f32x4*>(a.gn_c + (size_t)tk * a.K + ch);
                const int srow = min(tok0 / a.L + sm0 + u_s, m_last / a.L);                // the sample (a GroupNorm "row")
                mu[u] = a.gn_stats[((size_t)srow * 8 + grp) * 2]; rs[u] = a.gn_stats[((size_t)srow * 8 + grp) * 2 + 1];
                if (++u_i == n_it) { u_i = 0; ++u_s; }
              }
              float p1[NW], p2[NW];
#pragma unroll
              for (int u = 0; u < NW; ++u) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const float h = (hv[u][j] - mu[u]) * rs[u];
                  const float d = dv[u][j] * tw_mish_grad(h * gam[j] + bet[j]) * gam[j];
                  hv[u][j] = h; dv[u][j] = d;
                  s1 += d; s2 += d * h;
                }
                p1[u] = s1; p2[u] = s2;
              }
              // over the group's lanes (the low bits of the lane index below G4) and the row slots (the bits from LR up)
              for (int b = 1; b < G4; b <<= 1) {
#pragma unroll
                for (int u = 0; u < NW; ++u) { p1[u] += __shfl_xor(p1[u], b); p2[u] += __shfl_xor(p2[u], b); }
              }
              for (int b = LR; b < 64; b <<= 1) {
#pragma unroll
                for (int u = 0; u < NW; ++u) { p1[u] += __shfl_xor(p1[u], b); p2[u] += __shfl_xor(p2[u], b); }
              }
              // wave scratch [group of the chunk (<= 8)][instruction (<= 8)][2]: per-instruction sums -> per-sample totals at the sample's first instruction
              if (sr == 0 && sl == gl * G4) {
#pragma unroll
                for (int u = 0; u < NW; ++u) { wscr[(gl * 8 + u) * 2] = p1[u]; wscr[(gl * 8 + u) * 2 + 1] = p2[u]; }
              }
              tw_wave_sync();
              if (lane < n_gl * spp) {
                const int g2 = lane / spp, s2i = lane - g2 * spp;
                float t1 = 0.f, t2 = 0.f;
                for (int i2 = 0; i2 < n_it; ++i2) { t1 += wscr[(g2 * 8 + s2i * n_it + i2) * 2]; t2 += wscr[(g2 * 8 + s2i * n_it + i2) * 2 + 1]; }
                wscr[(g2 * 8 + s2i * n_it) * 2] = t1 * inv_cnt; wscr[(g2 * 8 + s2i * n_it) * 2 + 1] = t2 * inv_cnt;
              }
              tw_wave_sync();
              u_s = 0; u_i = 0;
#pragma unroll
              for (int u = 0; u < NW; ++u) {
                const float m1 = wscr[(gl * 8 + u_s * n_it) * 2], m2 = wscr[(gl * 8 + u_s * n_it) * 2 + 1];
                if (++u_i == n_it) { u_i = 0; ++u_s; }
                const int t = t_lo + u * RP + sr;
                const bool live = tok0 + t < a.M;                                            // (whole samples: M % L == 0)
                f32x4 x;
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = live ? (dv[u][j] - m1 - hv[u][j] * m2) * rs[u] : 0.f;
                amax = amax4(x, amax);
                unsigned h0, h1, l0, l1;
                split4(x * s_in, h0, h1, l0, l1);
                char* dst = smem + trow(t) * XROW + 8 * sl;
                *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(dst + 2 * KC) = u32x2{l0, l1};
              }
              tw_wave_sync();                               // (the scratch is reused by the next pass)
            }
          }
        }
        __syncthreads();

        // ---- D^T[c_out][token] += sum_tap W_tap X_shifted^T over this chunk's channels.  Step s = (tap, k16 step) = (s >> log2 KS, s & (KS - 1)).
        // Software-pipelined by hand: the weight fragments (global / L2 -> registers) and the token fragments (LDS) of step s + 1 are requested
        // before the MFMAs of step s (two register sets each, two steps unrolled so that every set has a static name); the second wave of the
        // SIMD (the CU's other block) covers what that leaves exposed
        const int n_steps = 5 * KS;
        const int lks = 31 - __builtin_clz(KS);
        const size_t wtap = (size_t)(a.N >> 5) * (a.K >> 4) * 2048;                         // bytes of one tap's planes
        const char* wbase = reinterpret_cast<const char*>(a.W) + ((size_t)(nb0 >> 5) * (a.K >> 4) + (size_t)kc * KS) * 2048 + lane * 16;
        const size_t wmb = (size_t)(a.K >> 4) * 2048;                                      // bytes between two 32-row blocks
        auto load_a = [&](u32x4 (&h)[MB], u32x4 (&l)[MB], int st) __attribute__((always_inline)) {
          const char* p = wbase + (size_t)(st >> lks) * wtap + (size_t)(st & (KS - 1)) * 2048;
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            h[mb] = *reinterpret_cast<const u32x4*>(p + mb * wmb);
            l[mb] = *reinterpret_cast<const u32x4*>(p + mb * wmb + 1024);
          }
        };
        auto load_b = [&](u32x4 (&h)[TW_NT], u32x4 (&l)[TW_NT], int st) __attribute__((always_inline)) {
          const int sh = a.dir * ((st >> lks) - 2) * XROW + 32 * (st & (KS - 1));
#pragma unroll
          for (int tg = 0; tg < TW_NT; ++tg) {
            const char* p = smem + boff[tg] + sh;
            h[tg] = *reinterpret_cast<const u32x4*>(p);
            l[tg] = *reinterpret_cast<const u32x4*>(p + 2 * KC);
          }
        };
        // one step: the 9 MB MFMAs of (ua, ub), and between them -- one request after each MFMA, pinned with sched_barrier: left alone hipcc
        // clusters every load of the loop body in front of its MFMAs, i.e. waits for them at once -- the fragments of step `nxt` into (na, nb)
        auto step = [&](const u32x4 (&uah)[MB], const u32x4 (&ual)[MB], const u32x4 (&ubh)[TW_NT], const u32x4 (&ubl)[TW_NT],
                        u32x4 (&nah)[MB], u32x4 (&nal)[MB], u32x4 (&nbh)[TW_NT], u32x4 (&nbl)[TW_NT], int nxt) __attribute__((always_inline)) {
          const int sc = min(nxt, n_steps - 1);                                            // (the last step re-requests its own fragments)
          const char* pa = wbase + (size_t)(sc >> lks) * wtap + (size_t)(sc & (KS - 1)) * 2048;
          const int shn = a.dir * ((sc >> lks) - 2) * XROW + 32 * (sc & (KS - 1));
          auto issue = [&](int q) __attribute__((always_inline)) {
            if (q < 2 * MB) {
              const u32x4 v = *reinterpret_cast<const u32x4*>(pa + (q >> 1) * wmb + (q & 1) * 1024);
              if (q & 1) nal[q >> 1] = v; else nah[q >> 1] = v;
            } else if (q < 2 * MB + 2 * TW_NT) {
              const int r = q - 2 * MB;
              const u32x4 v = *reinterpret_cast<const u32x4*>(smem + boff[r >> 1] + shn + (r & 1) * 2 * KC);
              if (r & 1) nbl[r >> 1] = v; else nbh[r >> 1] = v;
            }
          };
          int q = 0;
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int tg = 0; tg < TW_NT; ++tg) {
              f32x16 v = acc[mb][tg];
              v = mfma16(uah[mb], ubl[tg], v);
              issue(q++);
              __builtin_amdgcn_sched_barrier(0);
              v = mfma16(ual[mb], ubh[tg], v);
              issue(q++);
              __builtin_amdgcn_sched_barrier(0);
              v = mfma16(uah[mb], ubh[tg], v);
              issue(q++);
              __builtin_amdgcn_sched_barrier(0);
              acc[mb][tg] = v;
            }
        };
        u32x4 a0h[MB], a0l[MB], a1h[MB], a1l[MB], b0h[TW_NT], b0l[TW_NT], b1h[TW_NT], b1l[TW_NT];
        load_a(a0h, a0l, 0); load_b(b0h, b0l, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int st = (a.ablate & 1) ? n_steps : 0; st < n_steps; st += 2) {      // (n_steps = 5 KS is even)
          step(a0h, a0l, b0h, b0l, a1h, a1l, b1h, b1l, st + 1);
          step(a1h, a1l, b1h, b1l, a0h, a0l, b0h, b0l, st + 2);
        }
      }

      // ---- epilogue.  Accumulator layout of v_mfma_f32_32x32x16: lane (tok, kh), register 4 q + j = channel 32 mb + 8 q + 4 kh + j of token
      // 32 tg + tok: four consecutive channels per (q): one 16-byte access
      if (EPI == 0) {
#pragma unroll
        for (int tg = 0; tg < TW_NT; ++tg) {
          const int t = tok0 + 32 * tg + tok;
          const int tk = min(t, m_last);
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int n = nb0 + 32 * mb + 8 * q + 4 * kh;
              f32x4 v = quad(acc[mb][tg], q) * os;
              if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + n);
              if (a.resid) v += *reinterpret_cast<const f32x4*>(a.resid + (size_t)tk * a.ldr + n);
              if (a.resid2) v += *reinterpret_cast<const f32x4*>(a.resid2 + (size_t)tk * a.ldr2 + n);
              float* dst = n < a.N1 ? a.Y + (size_t)tk * a.ldy + n : a.Y2 + (size_t)tk * a.ldy2 + (n - a.N1);
              if (t < a.M && !(a.ablate & 4)) *reinterpret_cast<f32x4*>(dst) = v;
            }
        }
      } else {
        // c = acc os + bias (the stash); GroupNorm statistics per (sample, group): a group (N / 8 channels) is one 32-channel block
        // (N = 256) or half of one (N = 128: registers 0..7 / 8..15), always inside this wave's slice
        constexpr int NGW = 2;                               // groups of this wave's slice
        const int gsz = a.N >> 3;                            // channels of a group: 32 (MB = 2) or 16 (MB = 1)
        const int n_smp = TW_TB / a.L;
        const float inv_cnt = 1.f / (float)(a.L * gsz);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(prm + nb0 + 32 * mb + 8 * q + 4 * kh);
#pragma unroll
            for (int tg = 0; tg < TW_NT; ++tg)
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[mb][tg][4 * q + j] = acc[mb][tg][4 * q + j] * os + b4[j];
          }
        // the stash C (what the backward kernel normalises again)
#pragma unroll
        for (int tg = 0; tg < TW_NT; ++tg) {
          const int t = tok0 + 32 * tg + tok;
          const int tk = min(t, m_last);
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (t < a.M && !(a.ablate & 4)) *reinterpret_cast<f32x4*>(a.Cst + (size_t)tk * a.N + nb0 + 32 * mb + 8 * q + 4 * kh) = quad(acc[mb][tg], q);
        }
        // group gi of the wave: MB = 2 -> block gi, all 16 registers; MB = 1 -> block 0, registers 8 gi .. 8 gi + 7
        float mean_t[NGW][TW_NT], rstd_t[NGW][TW_NT];
        float my_mean = 0.f, my_var = 0.f;                   // of the (group, sample) this lane sums (lane < NGW n_smp)
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {                  // 0: mean, 1: variance about it (two passes like gn_fwd_kernel)
#pragma unroll
          for (int gi = 0; gi < NGW; ++gi)
#pragma unroll
            for (int tg = 0; tg < TW_NT; ++tg) {
              float s = 0.f;
#pragma unroll
              for (int i = 0; i < (MB == 2 ? 16 : 8); ++i) {
                const float c = MB == 2 ? acc[gi][tg][i] : acc[0][tg][8 * gi + i];
                const float d = rnd == 0 ? c : c - mean_t[gi][tg];
                s += rnd == 0 ? d : d * d;
              }
              s += __shfl_xor(s, 32);
              if (kh == 0) wscr[gi * TW_TB + 32 * tg + tok] = s;      // per-token sums of the group
            }
          tw_wave_sync();
          if (lane < NGW * n_smp) {                         // one lane per (group, sample): the sample's L tokens
            const int gi = lane / n_smp, sm = lane - gi * n_smp;
            float s = 0.f;
            for (int u = 0; u < a.L; ++u) s += wscr[gi * TW_TB + sm * a.L + u];
            s *= inv_cnt;
            wscr[192 + lane] = s;
            if (rnd == 0) my_mean = s; else my_var = s;
          }
          tw_wave_sync();
#pragma unroll
          for (int gi = 0; gi < NGW; ++gi)
#pragma unroll
            for (int tg = 0; tg < TW_NT; ++tg) {
              const float r = wscr[192 + gi * n_smp + divL(32 * tg + tok)];
              if (rnd == 0) mean_t[gi][tg] = r; else rstd_t[gi][tg] = 1.f / sqrtf(r + a.eps);
            }
          tw_wave_sync();
        }
        if (lane < NGW * n_smp) {                            // statistics for the backward pass: (sample, group) -> mean, rstd
          const int gi = lane / n_smp, sm = lane - gi * n_smp;
          if (tok0 + sm * a.L < a.M) {
            const int g = (nb0 >> 5) * (MB == 2 ? 1 : 2) + gi;                             // the group's index among the 8
            float* st = a.stats + ((size_t)(tok0 / a.L + sm) * 8 + g) * 2;
            st[0] = my_mean; st[1] = 1.f / sqrtf(my_var + a.eps);
          }
        }
        // y = mish((c - mean) rstd gamma + beta) + time bias + residual
#pragma unroll
        for (int tg = 0; tg < TW_NT; ++tg) {
          const int t = tok0 + 32 * tg + tok;
          const int tk = min(t, m_last);
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int n = nb0 + 32 * mb + 8 * q + 4 * kh;
              const int gi = MB == 2 ? mb : (q >> 1);
              const f32x4 gam = *reinterpret_cast<const f32x4*>(prm + 256 + n), bet = *reinterpret_cast<const f32x4*>(prm + 512 + n);
              f32x4 v;
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = tw_mish((acc[mb][tg][4 * q + j] - mean_t[gi][tg]) * rstd_t[gi][tg] * gam[j] + bet[j]);
              v += *reinterpret_cast<const f32x4*>(prm + 768 + n);
              if (a.resid) v += *reinterpret_cast<const f32x4*>(a.resid + (size_t)tk * a.ldr + n);
              if (t < a.M && !(a.ablate & 4)) *reinterpret_cast<f32x4*>(a.Y + (size_t)tk * a.ldy + n) = v;
            }
        }
      }
    }
  }

  amax = fmaxf(amax, __shfl_xor(amax, 32)); amax = fmaxf(amax, __shfl_xor(amax, 16)); amax = fmaxf(amax, __shfl_xor(amax, 8));
  amax = fmaxf(amax, __shfl_xor(amax, 4)); amax = fmaxf(amax, __shfl_xor(amax, 2)); amax = fmaxf(amax, __shfl_xor(amax, 1));
  record_amax_block_guarded<true>(a.amax_out, amax, reinterpret_cast<float*>(scr), a.range_flag, s_in, a.site);
}

// 256 registers per lane everywhere: two blocks share a CU (tw_geometry)
template <int PRO, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void tkw_kernel1(TkwArgs a, int n_tiles, int kc32, int n_chunks, int n_rows, int n_pass, int mulL) { tkw_body<1, PRO, EPI>(a, n_tiles, kc32, n_chunks, n_rows, n_pass, mulL); }
template <int PRO, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void tkw_kernel2(TkwArgs a, int n_tiles, int kc32, int n_chunks, int n_rows, int n_pass, int mulL) { tkw_body<2, PRO, EPI>(a, n_tiles, kc32, n_chunks, n_rows, n_pass, mulL); }

namespace {

// chunk of operand channels that fits the LDS with the padded tile (at most 256), the tile's rows and bytes
bool tw_geometry(int L, int N, int K, TwGeom* g) {
  if (L < 3 || TW_TB % L != 0) return false;
  if (!(N == 128 || N == 256 || N == 512)) return false;
  if (K < 32 || K > 512 || (K & (K - 1)) != 0) return false;      // powers of two only: the tap / chunk indexing shifts and masks by K / 32, KC / 32 (K = 96 indexed taps up to 7)
  const int rows = TW_TB + 2 * (TW_TB / L + 1);
  // chunks of at most 128 operand channels: the tile (rows x (4 KC + 16) bytes) then leaves room for TWO blocks per CU (2 x 78 KB), so that one
  // block's staging / epilogue (memory latency, GroupNorm arithmetic) runs beside the other's MFMA phase -- with one wave per SIMD nothing overlaps
  int KC = std::min(K, 128);
  while (KC > 32 && (size_t)rows * (4 * KC + 16) + TW_SCRATCH > 80 * 1024) KC >>= 1;
  if (KC < 32 || K % KC != 0) return false;
  g->KC = KC; g->n_chunks = K / KC; g->rows = rows; g->xrow = 4 * KC + 16;
  g->n_pass = N == 512 ? 2 : 1;
  g->lds = (size_t)rows * g->xrow + TW_SCRATCH;
  return true;
}

// the PRO template value of a GroupNorm-backward launch (0: not covered): the wave instructions of one staging pass = whole samples, at most 8
int tw_pro_variant(int L, int N, int K) {
  TwGeom g;
  if (!tw_geometry(L, N, K, &g)) return 0;
  const int RP = 64 / (g.KC / 4);
  if (L % RP != 0 || (K / 8) % 4 != 0) return 0;
  const int n_it = L / RP, n_smp = TW_TB / L;
  // every wave's share of samples must be whole passes: shares are floor((w + 1) n / 4) - floor(w n / 4)
  for (int nw : {6, 8, 4, 3}) {
    if (nw % n_it != 0) continue;
    const int spp = nw / n_it;
    bool ok = true;
    for (int w = 0; w < 4; ++w) ok &= ((((w + 1) * n_smp) >> 2) - ((w * n_smp) >> 2)) % spp == 0;
    if (ok && (LR_GROUPS_OK(g.KC, K))) return nw;
  }
  return 0;
}

template <int MB, int PRO, int EPI> int tkw_go(const TkwArgs& a, const TwGeom& g, hipStream_t s) {
  const int n_tiles = (a.M + TW_TB - 1) / TW_TB;
  const int per_cu = g.lds * 2 <= 160 * 1024 ? 2 : 1;        // (all variants compile to <= 256 registers: two waves per SIMD)
  const int nb = std::min(n_tiles, per_cu * device_cu_count());
  if (MB == 1) hipLaunchKernelGGL((tkw_kernel1<PRO, EPI>), dim3(nb), dim3(256), g.lds, s, a, n_tiles, g.KC >> 5, g.n_chunks, g.rows, g.n_pass, (65536 + a.L - 1) / a.L);
  else hipLaunchKernelGGL((tkw_kernel2<PRO, EPI>), dim3(nb), dim3(256), g.lds, s, a, n_tiles, g.KC >> 5, g.n_chunks, g.rows, g.n_pass, (65536 + a.L - 1) / a.L);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace

namespace { int tw_pro_variant(int L, int N, int K); }
bool tkw_applicable(int M, int L, int N, int K, int pro, int epi) {
  TwGeom g;
  if (!(M > 0 && M % L == 0 && tw_geometry(L, N, K, &g))) return false;
  if ((long)M * std::max(N, K) * 4 >= (1l << 32)) return false;      // (int offsets inside the kernel are size_t; kept as a sanity bound)
  if (pro) {      // GroupNorm backward: a wave stages whole samples, a sample's rows in whole wave instructions, all of them in registers at once
    if (tw_pro_variant(L, N, K) == 0) return false;
    if (g.n_chunks > 1 && (K / 8) > g.KC) return false;
  }
  if (epi && N > 256) return false;                                   // (no forward layer has 512 output channels; the epilogue's parameter table holds 256)
  return true;
}

int launch_tkw(const TkwArgs& a, hipStream_t s) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const int pro = a.gn_c ? 1 : 0, epi = a.Cst ? 1 : 0;
  TwGeom g;
  RAMP_REQUIRE(tkw_applicable(a.M, a.L, a.N, a.K, pro, epi) && tw_geometry(a.L, a.N, a.K, &g),
               "tkw: C_out in {128, 256, 512}, C_in a multiple of 32 up to 512, tokens per sample >= 3 dividing 96 (24 with the GroupNorm backward), whole samples");
  RAMP_REQUIRE(a.X && a.W && a.Y && (a.dir == 1 || a.dir == -1), "tkw: bad operand");
  RAMP_REQUIRE(a.K1 == a.K || (a.X2 && a.K1 > 0 && a.K1 < a.K && a.K1 % 4 == 0 && !pro), "tkw: bad operand split");
  RAMP_REQUIRE(a.N1 == a.N || (a.Y2 && a.N1 > 0 && a.N1 < a.N && a.N1 % 8 == 0 && !epi), "tkw: bad output split");
  RAMP_REQUIRE(al16(a.X) && al16(a.X2) && al16(a.W) && al16(a.Y) && al16(a.Y2) && al16(a.bias) && al16(a.resid) && al16(a.resid2) && al16(a.gn_c) &&
               al16(a.gn_gamma) && al16(a.gn_beta) && al16(a.Cst) && al16(a.gamma) && al16(a.beta) && al16(a.tbias) && a.ldx % 4 == 0 && a.ldx2 % 4 == 0 &&
               a.ldy % 4 == 0 && a.ldy2 % 4 == 0 && a.ldr % 4 == 0 && a.ldr2 % 4 == 0, "tkw: operands must be 16-byte aligned");
  if (pro) RAMP_REQUIRE(a.gn_stats && a.gn_gamma && a.gn_beta && a.ldx >= a.K, "tkw: GroupNorm-backward operand incomplete");
  if (epi) RAMP_REQUIRE(a.stats && a.gamma && a.beta && a.bias && !a.resid2 && a.ldy >= a.N, "tkw: GroupNorm epilogue incomplete");
  const size_t ybytes = ((size_t)(a.M - 1) * a.ldy + a.N1) * 4, xbytes = ((size_t)(a.M - 1) * a.ldx + a.K1) * 4;
  RAMP_REQUIRE(!ranges_overlap(a.Y, ybytes, a.X, xbytes), "tkw: the output must not overlap the operand");
  RAMP_REQUIRE(!epi || !ranges_overlap(a.Cst, (size_t)a.M * a.N * 4, a.X, xbytes), "tkw: the stash must not overlap the operand");
  const int mbv = a.N == 128 ? 1 : 2;
  const int prov = pro ? tw_pro_variant(a.L, a.N, a.K) : 0;
#define TW_CASE(MBV, PROV, EPIV) if (mbv == MBV && prov == PROV && epi == EPIV) return tkw_go<MBV, PROV, EPIV>(a, g, s);
  TW_CASE(1, 0, 0) TW_CASE(1, 0, 1) TW_CASE(1, 3, 0) TW_CASE(1, 4, 0) TW_CASE(1, 6, 0) TW_CASE(1, 8, 0)
  TW_CASE(2, 0, 0) TW_CASE(2, 0, 1) TW_CASE(2, 3, 0) TW_CASE(2, 4, 0) TW_CASE(2, 6, 0) TW_CASE(2, 8, 0)
#undef TW_CASE
  RAMP_REQUIRE(false, "tkw: variant not built");
}

int init_tkw_attributes() {
#define TW_ATTR(KV, PROV, EPIV) RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&KV<PROV, EPIV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
  TW_ATTR(tkw_kernel1, 0, 0); TW_ATTR(tkw_kernel1, 0, 1); TW_ATTR(tkw_kernel1, 3, 0); TW_ATTR(tkw_kernel1, 4, 0); TW_ATTR(tkw_kernel1, 6, 0); TW_ATTR(tkw_kernel1, 8, 0);
  TW_ATTR(tkw_kernel2, 0, 0); TW_ATTR(tkw_kernel2, 0, 1); TW_ATTR(tkw_kernel2, 3, 0); TW_ATTR(tkw_kernel2, 4, 0); TW_ATTR(tkw_kernel2, 6, 0); TW_ATTR(tkw_kernel2, 8, 0);
#undef TW_ATTR
  return 0;
}

}  // namespace ramp

// Self-attention of a transformer block fused with the linear layer behind it, as SAMPLE-OWNING waves
// (layers_attention_mini.py:101-127: softmax(q k^T / 8) v per head, :127 to_out, :132 x = attn1(norm1(x)) + x).
//
//   forward  (ato_kernel):  z1 = zin + Wo attention(q, k, v) + bo + cross-attention constant of the row's scene variant
//                           -- replaces attn2_fwd_kernel + the out-projection launch: o never reaches HBM.
//
// Dataflow.  A wave owns T = 48 (or 32) consecutive tokens = WHOLE samples of L tokens (L | T: 48, 24, 12, 6 at horizon 48),
// so the attention of its samples is wave-local -- no LDS exchange, no barrier inside it.  Everything is 16 x 16 tiles on
// v_mfma_f32_16x16x32_f16 / v_mfma_f32_16x16x16_f16 with the fp16x3 split (two scaled fp16 planes per operand, h1 h1' + h1 h2' +
// h2 h1', fp32 accumulate; the scales of the attention-internal operands are exact per-wave powers of two taken from the
// operands themselves, the scale of the linear layer's operand is the call site's delayed scale like every other GEMM).
// The 16 x 16 operand and accumulator layouts of these instructions are symmetric (operand: lane & 15 = row / column, lane >> 4
// and the register index = k; accumulator: lane & 15 = column, 4 (lane >> 4) + register = row), which is what makes the chain
// register-resident:
//   * q^T, k^T fragments [feature 4 g + i][token c] are loaded straight from the row-major qkv rows (16 bytes per lane) and ARE
//     the A / B operands of S^T = K Q^T (keys in registers, query on the lane: the softmax over keys is in-lane + two shuffles,
//     as in attention.hip); the whole T x T score matrix of the wave is computed and entries across samples are masked;
//   * P^T in its accumulator registers IS the B operand of O^T = V^T P^T (contraction over keys = the register index);
//     v is loaded with the token index in the registers (4 dwords) so that it is that product's A operand;
//   * O^T [feature 4 g + i][token c] IS the B operand of the output projection y^T = Wo o^T (the k order of the packed weight
//     fragments is permuted to the accumulator's row order at pack time, ato_pack);
//   * the projection's weights (256 KB of fp16 planes) stream through an LDS ring in head order, 32 KB per slab = half a head's
//     k range x 128 output features... (slab s = head s / 2, output features [128 (s & 1), +128)); 256 x T accumulators per wave.
// Tokens past M are clamped for loads (finite garbage that is never stored).
#include "common.h"
#include "tokmma.h"

#include <algorithm>

namespace ramp {

namespace {

typedef _Float16 half4v __attribute__((ext_vector_type(4)));

constexpr int AT_SLAB = 32 * 1024;                      // 8 output blocks of 16 x 2 k32 steps x 2 planes x 1 KB
constexpr int AT_R = 2;                                 // ring slots (v1: slab g + 1 is fetched while slab g computes)
constexpr size_t AT_LDS = (size_t)AT_R * AT_SLAB;

__device__ __forceinline__ f32x4 mm32(const u32x4 a, const u32x4 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mm16(const u32x2 a, const u32x2 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(half4v, a), __builtin_bit_cast(half4v, b), c, 0, 0, 0);
}
// 2^(target - floor(log2 mx)): mx lands in [2^target, 2^(target + 1)); 1 for mx == 0 / non-finite
__device__ __forceinline__ float pow2_scale(float mx, int target) {
  float s = 1.f;
  if (mx > 0.f && mx < 3.0e38f) {
    const int eb = (int)((__builtin_bit_cast(unsigned, mx) >> 23) & 0xffu);
    int sb = 254 + target - eb;
    sb = sb < 1 ? 1 : (sb > 254 ? 254 : sb);
    s = __builtin_bit_cast(float, (unsigned)sb << 23);
  }
  return s;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// four scaled floats -> two dwords of each plane: x = hi + lo (both RNE)
__device__ __forceinline__ void split4s(const f32x4 a, float s, u32x2& hi, u32x2& lo) {
  unsigned h0, h1, l0, l1;
  split4(a * s, h0, h1, l0, l1);
  hi = u32x2{h0, h1}; lo = u32x2{l0, l1};
}
__device__ __forceinline__ u32x4 cat2(const u32x2 a, const u32x2 b) { return u32x4{a[0], a[1], b[0], b[1]}; }

}  // namespace

// ---- weight stream of the output projection --------------------------------------------------------------------------------
// W [256 n][256 k] row-major -> out [slab 8][nbl 8][j 2][plane 2][lane 64][8 halves]: slab s serves head s >> 1 (k = 64 h ..) and
// output features 128 (s & 1) + 16 nbl + (lane & 15); the fragment of k32 step j holds, for lane group kq = lane >> 4, the k values
// 64 h + 32 j + {4 kq + e, e < 4} and 64 h + 32 j + 16 + {4 kq + e - 4, e >= 4}: the row order of two stacked accumulator tiles.
__global__ void ato_pack_kernel(const float* __restrict__ W, unsigned short* __restrict__ out, float scale) {
  const int idx = blockIdx.x * 256 + threadIdx.x;            // (s, nbl, j, lane): 8 * 8 * 2 * 64 = 8192
  if (idx >= 8192) return;
  const int lane = idx & 63, j = (idx >> 6) & 1, nbl = (idx >> 7) & 7, s = idx >> 10;
  const int h = s >> 1, n = 128 * (s & 1) + 16 * nbl + (lane & 15), kq = lane >> 4;
  _Float16 hi[8], lo[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int kl = e < 4 ? 4 * kq + e : 16 + 4 * kq + (e - 4);
    const float x = W[(long)n * 256 + 64 * h + 32 * j + kl] * scale;
    hi[e] = (_Float16)x;
    lo[e] = (_Float16)(x - (float)hi[e]);
  }
  unsigned short* o = out + ((long)((s * 8 + nbl) * 2 + j) * 2) * 512 + lane * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) { o[e] = __builtin_bit_cast(unsigned short, hi[e]); o[512 + e] = __builtin_bit_cast(unsigned short, lo[e]); }
}
int ato_pack(const float* W, float scale, unsigned short* out, hipStream_t s) {
  RAMP_REQUIRE(W && out, "ato_pack: null operand");
  hipLaunchKernelGGL(ato_pack_kernel, dim3(32), dim3(256), 0, s, W, out, scale);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

// NG: 16-token groups per wave (3: T = 48, 2: T = 32).  EPI bit 1: row-variant constant.
template <int NG, bool RB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void ato_kernel(AtoArgs a, int n_tiles) {
  constexpr int T = 16 * NG;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int n_my = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

  const float s_in = scale_of(a.amax_in);
  const float os = a.wsi / s_in;
  float amax = 0.f;

  // sample of each key row / query column of the wave's T x T score matrix (tile-independent: wave tiles start at sample starts)
  int samp_q[NG], samp_k[NG][4];
#pragma unroll
  for (int t = 0; t < NG; ++t) {
    samp_q[t] = (16 * t + c) / a.L;
#pragma unroll
    for (int i = 0; i < 4; ++i) samp_k[t][i] = (16 * t + 4 * g + i) / a.L;
  }

  // ---- weight ring (v1): wave w copies bytes [8 w KB, +8 KB) of a slab as 8 LDS-DMA pieces; slab g + 1 is issued behind the barrier
  // of slab g (every wave has left slab g - 1, whose slot it overwrites) and waited for, with everything else in flight, at the
  // top of slab g + 1.
  const char* wsrc = reinterpret_cast<const char*>(a.W) + wave * 8192 + lane * 16;
  int is_g = 0;
  auto issue_slab = [&]() __attribute__((always_inline)) {
    const char* src = wsrc + (long)(is_g & 7) * AT_SLAB;
    const unsigned dst = (unsigned)(uintptr_t)(smem + (is_g & (AT_R - 1)) * AT_SLAB + wave * 8192);
#define AT_PIECE(C) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2" \
                                 :: "v"(src + ((C) >> 2) * 4096), "s"(dst + ((C) >> 2) * 4096), "n"(((C) & 3) * 1024) : "memory", "m0")
    AT_PIECE(0); AT_PIECE(1); AT_PIECE(2); AT_PIECE(3); AT_PIECE(4); AT_PIECE(5); AT_PIECE(6); AT_PIECE(7);
#undef AT_PIECE
    ++is_g;
  };
  int gs = 0;                                               // slabs consumed
  const char* rd = smem + lane * 16;

  issue_slab();                                             // slab 0 of the first tile

  for (int ti = 0; ti < n_my; ++ti) {
    const int tile = (int)blockIdx.x + ti * (int)gridDim.x;
    const long tok0 = (long)tile * (4 * T) + wave * T;
    auto tok_of = [&](int tl) __attribute__((always_inline)) {      // token tl of the wave, clamped into [0, M)
      const long t = tok0 + tl;
      return t < a.M ? t : (long)a.M - 1;
    };

    f32x4 acc[16][NG];
#pragma unroll
    for (int nb = 0; nb < 16; ++nb)
#pragma unroll
      for (int t = 0; t < NG; ++t) acc[nb][t] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int h = 0; h < 4; ++h) {
      // ================= attention of head h on the wave's T tokens =================
      u32x4 oh[NG][2], ol[NG][2];                           // o^T planes: [token group][k32 step of the head]
      {
        // ---- q^T, k^T: lane (c, g) holds features 16 fb + 4 g + i of token 16 t + c
        f32x4 qr[4][NG], kr[4][NG];
#pragma unroll
        for (int t = 0; t < NG; ++t) {
          const float* row = a.QKV + tok_of(16 * t + c) * 768 + 64 * h + 4 * g;
#pragma unroll
          for (int fb = 0; fb < 4; ++fb) {
            qr[fb][t] = *reinterpret_cast<const f32x4*>(row + 16 * fb);
            kr[fb][t] = *reinterpret_cast<const f32x4*>(row + 256 + 16 * fb);
          }
        }
        // ---- v: lane (c, g) holds feature 16 fb + c of tokens 16 t + 4 g + i
        f32x4 vr[4][NG];
#pragma unroll
        for (int t = 0; t < NG; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float* row = a.QKV + tok_of(16 * t + 4 * g + i) * 768 + 512 + 64 * h + c;
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) vr[fb][t][i] = row[16 * fb];
          }
        float mq = 0.f, mk = 0.f, mv = 0.f;
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
          for (int t = 0; t < NG; ++t) { mq = amax4(qr[fb][t], mq); mk = amax4(kr[fb][t], mk); mv = amax4(vr[fb][t], mv); }
        mq = wave_max(mq); mk = wave_max(mk); mv = wave_max(mv);
        const float sq = pow2_scale(mq, 13), sk = pow2_scale(mk, 13), sv = pow2_scale(mv, 13);

        // ---- S^T = K Q^T: tile (kg, qg) over d = 64 = 2 k32 steps (fb pairs)
        f32x4 st[NG][NG];
        {
          u32x4 qh[NG][2], ql[NG][2], kh[NG][2], kl[NG][2];
#pragma unroll
          for (int t = 0; t < NG; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              u32x2 h0, l0, h1, l1;
              split4s(qr[2 * j][t], sq, h0, l0); split4s(qr[2 * j + 1][t], sq, h1, l1);
              qh[t][j] = cat2(h0, h1); ql[t][j] = cat2(l0, l1);
              split4s(kr[2 * j][t], sk, h0, l0); split4s(kr[2 * j + 1][t], sk, h1, l1);
              kh[t][j] = cat2(h0, h1); kl[t][j] = cat2(l0, l1);
            }
#pragma unroll
          for (int kg = 0; kg < NG; ++kg)
#pragma unroll
            for (int qg = 0; qg < NG; ++qg) {
              f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                s = mm32(kh[kg][j], ql[qg][j], s);
                s = mm32(kl[kg][j], qh[qg][j], s);
                s = mm32(kh[kg][j], qh[qg][j], s);
              }
              st[kg][qg] = s;
            }
        }
        // ---- P^T = softmax over keys (registers + lane groups) of S^T / 8, masked to the query's own sample
        const float ssc = 0.125f / (sq * sk);
#pragma unroll
        for (int qg = 0; qg < NG; ++qg) {
          float mx = -3.0e38f;
#pragma unroll
          for (int kg = 0; kg < NG; ++kg)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float v = samp_k[kg][i] == samp_q[qg] ? st[kg][qg][i] * ssc : -3.0e38f;
              st[kg][qg][i] = v;
              mx = fmaxf(mx, v);
            }
          mx = fmaxf(mx, __shfl_xor(mx, 16));
          mx = fmaxf(mx, __shfl_xor(mx, 32));
          float sum = 0.f;
#pragma unroll
          for (int kg = 0; kg < NG; ++kg)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float e = samp_k[kg][i] == samp_q[qg] ? expf(st[kg][qg][i] - mx) : 0.f;
              st[kg][qg][i] = e;
              sum += e;
            }
          sum += __shfl_xor(sum, 16);
          sum += __shfl_xor(sum, 32);
          const float inv = 8192.f / sum;                   // P^T scaled by 2^13 for its fp16 planes
#pragma unroll
          for (int kg = 0; kg < NG; ++kg) st[kg][qg] = st[kg][qg] * inv;
        }
        // ---- O^T = V^T P^T: tile (fb, qg), contraction over the keys: key groups (0, 1) as one k32 step, a third group as a k16 step
        f32x4 ot[4][NG];
        {
          u32x2 ph[NG][NG], pl[NG][NG], vh[4][NG], vl[4][NG];
#pragma unroll
          for (int kg = 0; kg < NG; ++kg)
#pragma unroll
            for (int qg = 0; qg < NG; ++qg) split4s(st[kg][qg], 1.f, ph[kg][qg], pl[kg][qg]);
#pragma unroll
          for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int kg = 0; kg < NG; ++kg) split4s(vr[fb][kg], sv, vh[fb][kg], vl[fb][kg]);
#pragma unroll
          for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int qg = 0; qg < NG; ++qg) {
              f32x4 o = {0.f, 0.f, 0.f, 0.f};
              const u32x4 vh01 = cat2(vh[fb][0], vh[fb][1]), vl01 = cat2(vl[fb][0], vl[fb][1]);
              const u32x4 ph01 = cat2(ph[0][qg], ph[1][qg]), pl01 = cat2(pl[0][qg], pl[1][qg]);
              o = mm32(vh01, pl01, o);
              o = mm32(vl01, ph01, o);
              o = mm32(vh01, ph01, o);
              if (NG == 3) {
                o = mm16(vh[fb][NG - 1], pl[NG - 1][qg], o);
                o = mm16(vl[fb][NG - 1], ph[NG - 1][qg], o);
                o = mm16(vh[fb][NG - 1], ph[NG - 1][qg], o);
              }
              ot[fb][qg] = o;
            }
        }
        // ---- o (true scale) -> recorded maximum, scaled planes of the projection's B operand
        const float so = 1.f / (sv * 8192.f);
#pragma unroll
        for (int t = 0; t < NG; ++t)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const f32x4 o0 = ot[2 * j][t] * so, o1 = ot[2 * j + 1][t] * so;
            amax = amax4(o0, amax); amax = amax4(o1, amax);
            u32x2 h0, l0, h1, l1;
            split4s(o0, s_in, h0, l0); split4s(o1, s_in, h1, l1);
            oh[t][j] = cat2(h0, h1); ol[t][j] = cat2(l0, l1);
          }
      }
      // ================= output projection: the head's two slabs (output features [0, 128), [128, 256)) =================
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my share of slab gs has landed (v1: everything else in flight too)
        __builtin_amdgcn_s_barrier();                       // slab gs certified; every wave has left slab gs - 1
        issue_slab();                                       // slab gs + 1 into the other slot (past the last tile: bytes nobody reads)
        const char* sl = rd + (gs & (AT_R - 1)) * AT_SLAB;
#pragma unroll
        for (int nbl = 0; nbl < 8; ++nbl)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const u32x4 wh = *reinterpret_cast<const u32x4*>(sl + ((nbl * 2 + j) * 2) * 1024);
            const u32x4 wl = *reinterpret_cast<const u32x4*>(sl + ((nbl * 2 + j) * 2 + 1) * 1024);
#pragma unroll
            for (int t = 0; t < NG; ++t) {
              f32x4 v = acc[8 * half + nbl][t];
              v = mm32(wh, ol[t][j], v);
              v = mm32(wl, oh[t][j], v);
              v = mm32(wh, oh[t][j], v);
              acc[8 * half + nbl][t] = v;
            }
          }
        ++gs;
      }
    }

    // ================= epilogue: y[tok][16 nb + 4 g ..] = acc * os + bias + constant of the row's variant + residual =================
    {
      long trow[NG]; bool live[NG]; int rbo[NG];
#pragma unroll
      for (int t = 0; t < NG; ++t) {
        const long tk = tok0 + 16 * t + c;
        live[t] = tk < a.M;
        trow[t] = live[t] ? tk : (long)a.M - 1;
        rbo[t] = RB ? a.rowvar[a.row0 + (int)(trow[t] / a.L)] * a.rb_stride : 0;
      }
#pragma unroll
      for (int nb = 0; nb < 16; ++nb) {
        const f32x4 bq = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + 16 * nb + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NG; ++t) {
          f32x4 v = acc[nb][t] * os + bq;
          if (RB) v += *reinterpret_cast<const f32x4*>(a.rowbias + rbo[t] + 16 * nb + 4 * g);
          v += *reinterpret_cast<const f32x4*>(a.resid + trow[t] * 256 + 16 * nb + 4 * g);
          if (live[t]) *reinterpret_cast<f32x4*>(a.Y + trow[t] * 256 + 16 * nb + 4 * g) = v;
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no LDS-DMA may outlive the block

  amax = wave_max(amax);
  if (lane == 0) {
    if (a.amax_out) atomicMax(reinterpret_cast<unsigned*>(a.amax_out), __builtin_bit_cast(unsigned, amax));
    if (a.range_flag && (!(amax * s_in < 60000.f) || (amax > 0.f && amax * s_in < 0.125f))) atomicMax(a.range_flag, a.site + 1);
  }
}

bool ato_applicable(int M, int L, int* ng) {
  int n = 0;
  if (L >= 1 && 48 % L == 0) n = 3; else if (L >= 1 && 32 % L == 0) n = 2;
  if (ng) *ng = n;
  return n != 0 && M > 0 && M % L == 0;
}

int launch_ato(const AtoArgs& a, hipStream_t s) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  int ng = 0;
  RAMP_REQUIRE(ato_applicable(a.M, a.L, &ng), "ato: tokens per sample must divide 48 or 32 (and M be whole samples)");
  RAMP_REQUIRE(a.QKV && a.W && a.resid && a.Y, "ato: null operand");
  RAMP_REQUIRE(al16(a.QKV) && al16(a.W) && al16(a.resid) && al16(a.Y) && al16(a.bias) && al16(a.rowbias), "ato: operands must be 16-byte aligned");
  RAMP_REQUIRE(!a.rowbias || (a.rowvar && a.rb_stride % 4 == 0), "ato: row-variant constant needs the row -> variant table");
  RAMP_REQUIRE(!ranges_overlap(a.Y, (size_t)a.M * 1024, a.resid, (size_t)a.M * 1024) && !ranges_overlap(a.Y, (size_t)a.M * 1024, a.QKV, (size_t)a.M * 3072),
               "ato: the output must not overlap the residual or qkv");
  const int T = 16 * ng, n_tiles = (a.M + 4 * T - 1) / (4 * T);
  const int nb = std::min(n_tiles, device_cu_count());
#define AT_GO(NGV, RBV) hipLaunchKernelGGL((ato_kernel<NGV, RBV>), dim3(nb), dim3(256), AT_LDS, s, a, n_tiles)
  if (ng == 3) { if (a.rowbias) AT_GO(3, true); else AT_GO(3, false); }
  else { if (a.rowbias) AT_GO(2, true); else AT_GO(2, false); }
#undef AT_GO
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

int init_atk_attributes() {
#define AT_ATTR(NGV, RBV) RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ato_kernel<NGV, RBV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AT_LDS))
  AT_ATTR(3, true); AT_ATTR(3, false); AT_ATTR(2, true); AT_ATTR(2, false);
#undef AT_ATTR
  return 0;
}

}  // namespace ramp

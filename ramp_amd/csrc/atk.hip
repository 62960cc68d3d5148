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
constexpr int AT_R = 3;                                 // ring slots: slab g + 2 is requested behind the barrier of slab g
constexpr int AT_BIAS = AT_R * AT_SLAB;                 // bias (256 floats)
constexpr int AT_RB = AT_BIAS + 256 * 4;                // row-variant constants (<= 4 x 256 floats)
constexpr int AT_V = AT_RB + 4 * 256 * 4;               // per wave: the v rows of one head, 4-row groups of 1 KB + 64 B pad (bank spread)
constexpr int AT_VG = 1024 + 64;                        // group stride
constexpr int AT_VW = 12 * AT_VG;                       // per wave (T = 48: 12 groups)
constexpr size_t AT_LDS = (size_t)AT_V + 4 * AT_VW;
static_assert(AT_LDS <= 160 * 1024, "LDS budget");

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

// NG: 16-token groups per wave (3: T = 48, 2: T = 32).  RB: per-row-variant constant.
//
// Schedule of a wave.  The work is a flat sequence of HEAD STEPS (tile, head): [attention of the head on registers] -> [slab 2 h] ->
// [slab 2 h + 1] -> (head 3: epilogue of the tile).  The raw q / k / v rows of the NEXT head step (the next tile's head 0 after head 3)
// are requested behind the first slab's barrier, so their latency hides behind 288 MFMAs and they occupy registers only while the
// projection runs; the v rows go through a wave-private LDS region by LDS-DMA behind the second barrier (the values vector instructions
// touch must fit the 256 architectural registers: the 192 accumulators live in the other half of the file).  Vector-memory order per
// head step: [DMA slab g + 2][q, k loads of the next step][DMA slab g + 3][v DMA of the next step](epilogue: 4 x (12 residual loads, 12 stores)); the
// head step starts with s_waitcnt vmcnt(0) (vmcnt(24) after an epilogue: its last batch stays in flight), which drains every LDS-DMA
// piece issued before -- each slab's pieces are issued two slabs ahead into a 3-slot ring, so the barrier at a slab top only has to
// publish them.
template <int NG, bool RB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void ato_kernel(AtoArgs a, int n_tiles) {
  constexpr int T = 16 * NG;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int n_my = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int n_steps = 4 * n_my;

  const float s_in = scale_of(a.amax_in);
  const float os = a.wsi / s_in;
  float amax = 0.f;

  // tables -> LDS (published by the first slab barrier)
  float* bs = reinterpret_cast<float*>(smem + AT_BIAS);
  bs[tid] = a.bias ? a.bias[tid] : 0.f;
  if (RB) {
    float* rbs = reinterpret_cast<float*>(smem + AT_RB);
    for (int v = 0; v < a.n_var; ++v) rbs[v * 256 + tid] = a.rowbias[(long)v * a.rb_stride + tid];
  }

  // sample of each key row / query column of the wave's T x T score matrix (tile-independent: wave tiles start at sample starts)
  int samp_q[NG], samp_k[NG][4];
#pragma unroll
  for (int t = 0; t < NG; ++t) {
    samp_q[t] = (16 * t + c) / a.L;
#pragma unroll
    for (int i = 0; i < 4; ++i) samp_k[t][i] = (16 * t + 4 * g + i) / a.L;
  }

  // ---- weight ring: wave w copies bytes [8 w KB, +8 KB) of a slab as 8 LDS-DMA pieces of 1 KB (inline asm, as ffx.hip / tkl.hip)
  const char* wsrc = reinterpret_cast<const char*>(a.W) + wave * 8192 + lane * 16;
  int is_g = 0;
  auto issue_slab = [&]() __attribute__((always_inline)) {
    const char* src = wsrc + (long)(is_g & 7) * AT_SLAB;
    const unsigned dst = (unsigned)(uintptr_t)(smem + (is_g % AT_R) * AT_SLAB + wave * 8192);
#define AT_PIECE(C) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2" \
                                 :: "v"(src + ((C) >> 2) * 4096), "s"(dst + ((C) >> 2) * 4096), "n"(((C) & 3) * 1024) : "memory", "m0")
    AT_PIECE(0); AT_PIECE(1); AT_PIECE(2); AT_PIECE(3); AT_PIECE(4); AT_PIECE(5); AT_PIECE(6); AT_PIECE(7);
#undef AT_PIECE
    ++is_g;
  };
  int gs = 0;                                               // slabs consumed
  const char* rd = smem + lane * 16;

  // byte offsets of the wave's rows in qkv (32-bit: launch_ato bounds M), clamped into [0, M): tokens past M read row M - 1
  const char* qbase = reinterpret_cast<const char*>(a.QKV);
  auto qk_off = [&](int tile, int t) __attribute__((always_inline)) {        // row of token 16 t + c, + 16 g bytes: features 4 g ..
    long tk = (long)tile * (4 * T) + wave * T + 16 * t + c;
    tk = tk < a.M ? tk : (long)a.M - 1;
    return (unsigned)(tk * 3072 + 16 * g);
  };
  f32x4 qr[4][NG], kr[4][NG];                               // raw q, k rows of the NEXT head step
  auto load_qk = [&](int tile, int h) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < NG; ++t) {
      const char* row = qbase + qk_off(tile, t) + 256 * h;
#pragma unroll
      for (int fb = 0; fb < 4; ++fb) {
        qr[fb][t] = *reinterpret_cast<const f32x4*>(row + 64 * fb);
        kr[fb][t] = *reinterpret_cast<const f32x4*>(row + 1024 + 64 * fb);
      }
    }
  };
  // v rows of a head: LDS-DMA into the wave's own LDS region (no register, no vector instruction): one instruction moves the 256-byte
  // head slices of 4 consecutive tokens (lane -> row lane >> 4, 16-byte chunk lane & 15); the attention reads them back with the token
  // index in the registers (4 ds_read_b32 per tile: the A operand of O^T = V^T P^T); groups are padded by 64 bytes: conflict-free
  const unsigned vdst = (unsigned)(uintptr_t)(smem + AT_V + wave * AT_VW);
  auto dma_v = [&](int tile, int h) __attribute__((always_inline)) {
#pragma unroll
    for (int gr = 0; gr < T / 4; ++gr) {
      long tk = (long)tile * (4 * T) + wave * T + 4 * gr + g;
      tk = tk < a.M ? tk : (long)a.M - 1;
      const char* src = qbase + (unsigned)(tk * 3072 + 2048 + 16 * c) + 256 * h;
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(vdst + gr * AT_VG) : "memory", "m0");
    }
  };
  const char* vrd = smem + AT_V + wave * AT_VW + g * AT_VG + 4 * c;      // + (4 kg) groups, + 256 i, + 64 fb

  issue_slab(); issue_slab();                               // slabs 0, 1 of the first tile
  load_qk((int)blockIdx.x, 0); dma_v((int)blockIdx.x, 0);

  // (the accumulators are cleared from ONE register the compiler cannot see through: as literal zeros hipcc hoists 192 zero registers
  // out of the head-step loop, spills them and reloads them in every iteration)
  f32x4 acc[16][NG];
  {
    float z = 0.f;
    asm volatile("" : "+v"(z));
#pragma unroll
    for (int nb = 0; nb < 16; ++nb)
#pragma unroll
      for (int t = 0; t < NG; ++t) acc[nb][t] = f32x4{z, z, z, z};
  }

#pragma unroll 1
  for (int hs = 0; hs < n_steps; ++hs) {
    const int h = hs & 3;
    const int tile = (int)blockIdx.x + (hs >> 2) * (int)gridDim.x;
    const int hs_n = hs + 1 < n_steps ? hs + 1 : hs;        // (the last step re-requests its own rows: unused)
    const int tile_n = (int)blockIdx.x + (hs_n >> 2) * (int)gridDim.x, h_n = hs_n & 3;
    const long tok0 = (long)tile * (4 * T) + wave * T;
    const bool full = tok0 + T <= a.M;                      // wave-uniform
    // the projection's accumulators belong in the accumulation half of the register file: left to itself hipcc keeps these loop-carried
    // values in architectural registers (copying them to AGPR temporaries around every MFMA) and spills the attention's operands
#pragma unroll
    for (int nb = 0; nb < 16; ++nb)
#pragma unroll
      for (int t = 0; t < NG; ++t) asm volatile("" : "+a"(acc[nb][t]));
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
      for (int t = 0; t < NG; ++t) { asm volatile("" : "+v"(qr[fb][t])); asm volatile("" : "+v"(kr[fb][t])); }

    // every request of the previous head step has landed (after an epilogue its last batch of 12 loads + 12 stores may stay in flight)
    if (h == 0 && hs > 0) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ================= attention of head h on the wave's T tokens =================
    // (ordered so that the values vector instructions touch stay below the 256 architectural registers: K planes 48, one query
    // group's Q planes 16 and score tiles 12 at a time, P planes 36, raw v 48; sched_barriers keep hipcc from merging the phases)
    u32x4 oh[NG][2], ol[NG][2];                             // o^T planes: [token group][k32 step of the head]
    {
      float mq = 0.f, mk = 0.f;
#pragma unroll
      for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int t = 0; t < NG; ++t) { mq = amax4(qr[fb][t], mq); mk = amax4(kr[fb][t], mk); }
      mq = wave_max(mq); mk = wave_max(mk);
      const float sq = pow2_scale(mq, 13), sk = pow2_scale(mk, 13);
      // exp2 of log2(e)-scaled logits; S^T / 8 with the operand scales undone
      const float ssc = 0.125f * 1.4426950408889634f / (sq * sk);

      u32x4 kh[NG][2], kl[NG][2];
#pragma unroll
      for (int t = 0; t < NG; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          u32x2 h0, l0, h1, l1;
          split4s(kr[2 * j][t], sk, h0, l0); split4s(kr[2 * j + 1][t], sk, h1, l1);
          kh[t][j] = cat2(h0, h1); kl[t][j] = cat2(l0, l1);
        }
      __builtin_amdgcn_sched_barrier(0);

      // ---- per query group: S^T = K Q^T (tiles (kg, qg), d = 64 = 2 k32 steps), P^T = softmax over keys (registers + lane groups),
      // masked to the query's own sample, scaled by 2^13 and split into its planes
      u32x2 ph[NG][NG], pl[NG][NG];
#pragma unroll
      for (int qg = 0; qg < NG; ++qg) {
        u32x4 qh[2], ql[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          u32x2 h0, l0, h1, l1;
          split4s(qr[2 * j][qg], sq, h0, l0); split4s(qr[2 * j + 1][qg], sq, h1, l1);
          qh[j] = cat2(h0, h1); ql[j] = cat2(l0, l1);
        }
        f32x4 st[NG];
#pragma unroll
        for (int kg = 0; kg < NG; ++kg) {
          f32x4 sv4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            sv4 = mm32(kh[kg][j], ql[j], sv4);
            sv4 = mm32(kl[kg][j], qh[j], sv4);
            sv4 = mm32(kh[kg][j], qh[j], sv4);
          }
          st[kg] = sv4;
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int kg = 0; kg < NG; ++kg)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = samp_k[kg][i] == samp_q[qg] ? st[kg][i] * ssc : -3.0e38f;
            st[kg][i] = v;
            mx = fmaxf(mx, v);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kg = 0; kg < NG; ++kg)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float e = __builtin_amdgcn_exp2f(st[kg][i] - mx);          // (masked entries: exp2(-3e38) = 0)
            st[kg][i] = e;
            sum += e;
          }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 8192.f * __builtin_amdgcn_rcpf(sum);               // (the sum's rcp: 1 ulp, a common factor of the column)
#pragma unroll
        for (int kg = 0; kg < NG; ++kg) { unsigned h0, h1, l0, l1; split4(st[kg] * inv, h0, h1, l0, l1); ph[kg][qg] = u32x2{h0, h1}; pl[kg][qg] = u32x2{l0, l1}; }
        __builtin_amdgcn_sched_barrier(0);
      }

      // ---- O^T = V^T P^T: tile (fb, qg), contraction over the keys: key groups (0, 1) as one k32 step, a third group as a k16 step
      float mv = 0.f;
      {
        f32x4 vt[NG];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
#pragma unroll
          for (int kg = 0; kg < NG; ++kg)
#pragma unroll
            for (int i = 0; i < 4; ++i) vt[kg][i] = *reinterpret_cast<const float*>(vrd + 4 * kg * AT_VG + 256 * i + 64 * fb);
#pragma unroll
          for (int kg = 0; kg < NG; ++kg) mv = amax4(vt[kg], mv);
        }
      }
      mv = wave_max(mv);
      const float sv = pow2_scale(mv, 13);
      const float so = 1.f / (sv * 8192.f);
      float lm[NG];
#pragma unroll
      for (int t = 0; t < NG; ++t) lm[t] = (full || tok0 + 16 * t + c < a.M) ? 1.f : 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {                         // feature-block pair (2 j, 2 j + 1) = the projection's k32 step j of this head
        u32x2 vh[2][NG], vl[2][NG];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int kg = 0; kg < NG; ++kg) {
            f32x4 vt;
#pragma unroll
            for (int i = 0; i < 4; ++i) vt[i] = *reinterpret_cast<const float*>(vrd + 4 * kg * AT_VG + 256 * i + 64 * (2 * j + f));
            split4s(vt, sv, vh[f][kg], vl[f][kg]);
          }
#pragma unroll
        for (int qg = 0; qg < NG; ++qg) {
          f32x4 o2[2];
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            const u32x4 vh01 = cat2(vh[f][0], vh[f][1]), vl01 = cat2(vl[f][0], vl[f][1]);
            const u32x4 ph01 = cat2(ph[0][qg], ph[1][qg]), pl01 = cat2(pl[0][qg], pl[1][qg]);
            o = mm32(vh01, pl01, o);
            o = mm32(vl01, ph01, o);
            o = mm32(vh01, ph01, o);
            if (NG == 3) {      // (a k32 step with an empty upper half, not v_mfma_f32_16x16x16_f16: behind that instruction hipcc placed the
                                // first read of the accumulator too early on gfx950 -- registers 0, 1 of the tile still held the previous sum)
              const u32x2 z2 = {0u, 0u};
              const u32x4 vh2 = cat2(vh[f][NG - 1], z2), vl2 = cat2(vl[f][NG - 1], z2);
              const u32x4 ph2 = cat2(ph[NG - 1][qg], z2), pl2 = cat2(pl[NG - 1][qg], z2);
              o = mm32(vh2, pl2, o);
              o = mm32(vl2, ph2, o);
              o = mm32(vh2, ph2, o);
            }
            o2[f] = o * so;                                 // o, true scale
          }
          // recorded maximum (live tokens only), scaled planes of the projection's B operand
          amax = amax4(o2[0] * lm[qg], amax); amax = amax4(o2[1] * lm[qg], amax);
          u32x2 h0, l0, h1, l1;
          split4s(o2[0], s_in, h0, l0); split4s(o2[1], s_in, h1, l1);
          oh[qg][j] = cat2(h0, h1); ol[qg][j] = cat2(l0, l1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ================= output projection: the head's two slabs (output features [0, 128), [128, 256)) =================
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __builtin_amdgcn_s_barrier();                         // slab gs is complete in LDS (every wave drained its pieces at the head-step start); slab gs - 1 is left
      issue_slab();                                         // slab gs + 2 into the slot of slab gs - 1 (past the last tile: bytes nobody reads)
      if (half == 0) load_qk(tile_n, h_n); else dma_v(tile_n, h_n);      // (this head's v was consumed before its first slab)
      const char* sl = rd + (gs % AT_R) * AT_SLAB;
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
          // (fragment reads may run two macro-steps ahead of their MFMAs, not a whole slab: 128 registers)
          if ((nbl * 2 + j) & 1) __builtin_amdgcn_sched_barrier(0);
        }
      ++gs;
    }

    // ================= epilogue (head 3): y[tok][16 nb + 4 g ..] = acc * os + bias + constant of the row's variant + residual =================
    if (h == 3) {
      const float* bsr = reinterpret_cast<const float*>(smem + AT_BIAS) + 4 * g;
      unsigned yoff[NG]; int rbo[NG];
#pragma unroll
      for (int t = 0; t < NG; ++t) {
        long tk = tok0 + 16 * t + c;
        tk = tk < a.M ? tk : (long)a.M - 1;
        yoff[t] = (unsigned)(tk * 1024 + 16 * g);
        rbo[t] = RB ? a.rowvar[a.row0 + (int)(tk / a.L)] * 256 + 4 * g : 0;
      }
      const char* rbase = reinterpret_cast<const char*>(a.resid);
      char* ybase = reinterpret_cast<char*>(a.Y);
      float z = 0.f;
      asm volatile("" : "+v"(z));
      if (full) {
        // all loads of a batch first, every store unconditional (a store behind a per-lane predicate sits in its own basic block behind
        // s_waitcnt vmcnt(0): DESIGN.md section 5); 8 batches x (6 loads, 6 stores) -- the head-step start counts on >= 24 operations here
#pragma unroll
        for (int b2 = 0; b2 < 8; ++b2) {
          f32x4 rz[2][NG];
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int t = 0; t < NG; ++t) rz[q][t] = *reinterpret_cast<const f32x4*>(rbase + yoff[t] + 64 * (2 * b2 + q));
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int nb = 2 * b2 + q;
            const f32x4 bq = *reinterpret_cast<const f32x4*>(bsr + 16 * nb);
#pragma unroll
            for (int t = 0; t < NG; ++t) {
              f32x4 v = acc[nb][t] * os + bq + rz[q][t];
              if (RB) v += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(smem + AT_RB) + rbo[t] + 16 * nb);
              *reinterpret_cast<f32x4*>(ybase + yoff[t] + 64 * nb) = v;
              acc[nb][t] = f32x4{z, z, z, z};
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        // the block's last, partly (or wholly) empty wave tile: per-token predicate
#pragma unroll
        for (int nb = 0; nb < 16; ++nb) {
          const f32x4 bq = *reinterpret_cast<const f32x4*>(bsr + 16 * nb);
#pragma unroll
          for (int t = 0; t < NG; ++t) {
            f32x4 v = acc[nb][t] * os + bq + *reinterpret_cast<const f32x4*>(rbase + yoff[t] + 64 * nb);
            if (RB) v += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(smem + AT_RB) + rbo[t] + 16 * nb);
            if (tok0 + 16 * t + c < a.M) *reinterpret_cast<f32x4*>(ybase + yoff[t] + 64 * nb) = v;
            acc[nb][t] = f32x4{z, z, z, z};
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the next head step's vmcnt(24) assumes a full epilogue's operation count)
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
  RAMP_REQUIRE(!a.rowbias || (a.rowvar && a.n_var >= 1 && a.n_var <= 4), "ato: row-variant constant needs the row -> variant table and 1 .. 4 variants");
  RAMP_REQUIRE((long)a.M * 3072 < (1l << 32), "ato: 32-bit row offsets bound M to 1398100 tokens");
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

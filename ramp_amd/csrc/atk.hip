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
#include "args_attention.h"
#include "tokmma.h"
#include "atkmma.h"

#include <algorithm>

namespace ramp {

namespace {

constexpr int AT_SLAB = 16 * 1024;                      // 4 output blocks of 16 x 2 k32 steps x 2 planes x 1 KB
constexpr int AT_R = 3;                                 // ring slots: slab g + 2 is requested during slab g
constexpr int AT_BIAS = AT_R * AT_SLAB;                 // bias (256 floats)
constexpr int AT_RB = AT_BIAS + 256 * 4;                // row-variant constants (<= 4 x 256 floats)
constexpr int AT_V = AT_RB + 4 * 256 * 4;               // per wave: the v rows of one head
constexpr int AT_K = AT_V + 4 * AT_VW;                  // per wave: the k rows of one head
constexpr size_t AT_LDS = (size_t)AT_K + 4 * AT_VW;
static_assert(AT_LDS <= 160 * 1024, "LDS budget");


}  // namespace

// ---- weight stream of the output projection --------------------------------------------------------------------------------
// W [256 n][256 k] row-major -> out [slab 8][nbl 8][j 2][plane 2][lane 64][8 halves] (= 16 ring slabs of 16 KB): slab s serves head s >> 1 (k = 64 h ..) and
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
// Schedule of a wave.  The work is a flat sequence of HEAD STEPS (tile, head): [attention of the head] -> [4 slabs of the projection: 16 output
// features x 4, the head's 64 k] -> (head 3: epilogue of the tile).  Register budget: the 192 accumulators live in the accumulation half of
// the file; what vector instructions touch must fit the 256 architectural registers (and hipcc spills long before that), so of the NEXT
// head step only the raw q rows wait in registers (48): its k and v rows go to wave-private LDS regions by LDS-DMA, in single pieces
// spread over the whole head step (the kernel is bound by the rate at which HBM requests can be issued: ~11 bytes per cycle and CU).
// Vector-memory order of a head step: wait(0) | S^T / softmax: the NEXT step's k pieces (4 per query group) | its q loads (12) | slab k:
// ring slab g + 2 + k (4 pieces, every other macro-step), v pieces 3 k .. 3 k + 2 | (epilogue).  (Round 4 touched one dword of every line of the
// epilogue's residual rows during slab 3 of every head step, to find them in the L2; round 5's counters showed the launch reading 8.3 KB per token where
// 4 KB are needed, and without the touches it is 10-18 % faster: removed.)  A slab's pieces are issued two slabs ahead into a 3-slot ring; slabs 2, 3 wait with vmcnt(6) (8 and more requests are
// younger than the slab they wait for), the wait at the head step's start drains everything.
// ABL (diagnostic twins, ramp_bench_gemm only): bit 0 per-wave s_memtime sums of a head step's phases; bit 1 no k / v LDS-DMA, bit 2 no
// ring LDS-DMA inside the slabs (wrong results)
template <int NG, bool RB, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void ato_kernel(AtoArgs a, int n_tiles) {
  constexpr int T = 16 * NG;
  constexpr bool STAMP = (ABL & 1) != 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int n_my = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int n_steps = 4 * n_my;

  const float s_in = scale_of(a.amax_in);
  const float os = a.wsi / s_in;
  float amax = 0.f;
  unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
  const unsigned long long t_start = STAMP ? __builtin_amdgcn_s_memtime() : 0, r_start = STAMP ? __builtin_amdgcn_s_memrealtime() : 0;
  auto stamp = [&](int k) __attribute__((always_inline)) {
    if (STAMP) { const unsigned long long t = __builtin_amdgcn_s_memtime(); if (tlast) tk[k] += t - tlast; tlast = t; }
  };

  // tables -> LDS (published by the first slab barrier)
  float* bs = reinterpret_cast<float*>(smem + AT_BIAS);
  bs[tid] = a.bias ? a.bias[tid] : 0.f;
  if (RB) {
    float* rbs = reinterpret_cast<float*>(smem + AT_RB);
    for (int v = 0; v < a.n_var; ++v) rbs[v * 256 + tid] = a.rowbias[(long)v * a.rb_stride + tid];
  }

  // which keys (rows 16 kg + 4 g + i of the wave's T x T score matrix) share the sample of query 16 qg + c: one bit per (kg, i), a
  // register per query group (tile-independent: wave tiles start at sample starts)
  unsigned kmask[NG];
#pragma unroll
  for (int qg = 0; qg < NG; ++qg) {
    unsigned m = 0;
#pragma unroll
    for (int kg = 0; kg < NG; ++kg)
#pragma unroll
      for (int i = 0; i < 4; ++i) m |= ((16 * kg + 4 * g + i) / a.L == (16 * qg + c) / a.L ? 1u : 0u) << (4 * kg + i);
    kmask[qg] = m;
  }

  // ---- weight ring: 16 KB slabs (4 output blocks of 16 x 2 k32 steps x 2 planes); wave w copies bytes [4 w KB, +4 KB) of a slab as 4
  // LDS-DMA pieces of 1 KB (inline asm, as ffx.hip / tkl.hip; scalar base + one 32-bit lane offset)
  const unsigned lane_w = (unsigned)(wave * 4096 + lane * 16);
  int is_g = 0;
  const char* ring_src = nullptr; unsigned ring_dst = 0;
  auto ring_begin = [&]() __attribute__((always_inline)) {
    ring_src = reinterpret_cast<const char*>(a.W) + (long)(is_g & 15) * AT_SLAB;      // scalar
    ring_dst = (unsigned)(uintptr_t)(smem + (is_g % AT_R) * AT_SLAB + wave * 4096);
    ++is_g;
  };
#define AT_PIECE(C) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" \
                                 :: "v"(lane_w), "s"(ring_src), "s"(ring_dst), "n"((C) * 1024) : "memory", "m0")
  auto ring_piece = [&](int cpc) __attribute__((always_inline)) {
    switch (cpc) { case 0: AT_PIECE(0); break; case 1: AT_PIECE(1); break; case 2: AT_PIECE(2); break; default: AT_PIECE(3); break; }
  };
  int gs = 0;                                               // slabs consumed
  const char* rd = smem + lane * 16;

  // rows of the wave in qkv: 32-bit byte offsets (launch_ato bounds M) from ONE per-use token index the compiler cannot see through
  // (left visible, hipcc hoists fifteen tile-independent partial sums out of the head-step loop and spills them), clamped into
  // [0, M): tokens past M read row M - 1
  const int m_last = a.M - 1;
  f32x4 qr[4][NG];                                          // raw q rows of the NEXT head step: lane (c, g) holds features 16 fb + 4 g + i of token 16 t + c
  auto load_q = [&](int tile, int h) __attribute__((always_inline)) {
    int tk0 = tile * (4 * T) + wave * T + c;
    asm volatile("" : "+v"(tk0));
    const char* qb = reinterpret_cast<const char*>(a.QKV) + 256 * h;       // scalar
#pragma unroll
    for (int t = 0; t < NG; ++t) {
      const char* row = qb + ((unsigned)min(tk0 + 16 * t, m_last) * 3072u + 16u * (unsigned)g);
#pragma unroll
      for (int fb = 0; fb < 4; ++fb) qr[fb][t] = *reinterpret_cast<const f32x4*>(row + 64 * fb);
    }
  };
  // k and v rows of a head: LDS-DMA into the wave's own LDS regions (no register, no vector instruction), one instruction = the 256-byte head
  // slices of 4 consecutive tokens (a "group": 1 KB + 64 bytes of padding, which spreads the groups over the banks).
  //   v: lane -> row lane >> 4, chunk lane & 15 ([row][chunk]); read back with the token index in the registers (4 ds_read_b32 per tile: the A
  //      operand of O^T = V^T P^T, lane (c, g) = feature 16 fb + c of tokens 16 kg + 4 g + i);
  //   k: lane -> row lane & 3, chunk lane >> 2 ([chunk][row]); read back as the A operand of S^T = K Q^T with ds_read_b128 (lane (c, g) =
  //      features 16 fb + 4 g .. of token 16 kg + c): the 16 lanes of a pass read 4 groups x 64 contiguous bytes, all 64 banks once.
  const unsigned vdst = (unsigned)(uintptr_t)(smem + AT_V + wave * AT_VW), kdst = (unsigned)(uintptr_t)(smem + AT_K + wave * AT_VW);
  int kv_tile = 0, kv_h = 0;                                // the head step the pieces being issued belong to
  auto dma_v_piece = [&](int gr) __attribute__((always_inline)) {
    int tk0 = kv_tile * (4 * T) + wave * T + g;             // token 4 gr + g; + 16 c bytes
    asm volatile("" : "+v"(tk0));
    const char* vb = reinterpret_cast<const char*>(a.QKV) + 2048 + 256 * kv_h;      // scalar
    const unsigned off = (unsigned)min(tk0 + 4 * gr, m_last) * 3072u + 16u * (unsigned)c;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off), "s"(vb), "s"(vdst + gr * AT_VG) : "memory", "m0");
  };
  auto dma_k_piece = [&](int gr) __attribute__((always_inline)) {
    int tk0 = kv_tile * (4 * T) + wave * T + (c & 3);       // token 4 gr + (lane & 3); chunk lane >> 2 = 4 g + (c >> 2)
    asm volatile("" : "+v"(tk0));
    const char* kb = reinterpret_cast<const char*>(a.QKV) + 1024 + 256 * kv_h;      // scalar
    const unsigned off = (unsigned)min(tk0 + 4 * gr, m_last) * 3072u + 16u * (unsigned)(4 * g + (c >> 2));
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off), "s"(kb), "s"(kdst + gr * AT_VG) : "memory", "m0");
  };
  const char* vrd = smem + AT_V + wave * AT_VW + g * AT_VG + 4 * c;                          // + (4 kg) groups, + 256 i, + 64 fb
  const char* krd = smem + AT_K + wave * AT_VW + (c >> 2) * AT_VG + 64 * g + 16 * (c & 3);  // + (4 kg) groups, + 256 fb

  // prologue: slabs 0, 1 of the first tile, the first head step's rows
  ring_begin(); ring_piece(0); ring_piece(1); ring_piece(2); ring_piece(3);
  ring_begin(); ring_piece(0); ring_piece(1); ring_piece(2); ring_piece(3);
  kv_tile = (int)blockIdx.x; kv_h = 0;
#pragma unroll
  for (int gr = 0; gr < T / 4; ++gr) { dma_k_piece(gr); dma_v_piece(gr); }
  load_q((int)blockIdx.x, 0);

  f32x4 acc[16][NG];                                        // the projection's accumulators

#pragma unroll 1
  for (int ti = 0; ti < n_my; ++ti) {
  const int tile = (int)blockIdx.x + ti * (int)gridDim.x;
  const long tok0 = (long)tile * (4 * T) + wave * T;
  const bool full = tok0 + T <= a.M;                        // wave-uniform
  // (cleared by inline asm: literal zeros make hipcc hoist 192 zero registers out of the loops and spill them; the accumulators are
  // defined here and die in the epilogue below, so nothing of them is carried around the tile loop)
#pragma unroll
  for (int nb = 0; nb < 16; ++nb)
#pragma unroll
    for (int t = 0; t < NG; ++t) {
      float z0, z1, z2, z3;
      asm volatile("v_accvgpr_write_b32 %0, 0\n\tv_accvgpr_write_b32 %1, 0\n\tv_accvgpr_write_b32 %2, 0\n\tv_accvgpr_write_b32 %3, 0" : "=a"(z0), "=a"(z1), "=a"(z2), "=a"(z3));
      acc[nb][t] = f32x4{z0, z1, z2, z3};
    }
#pragma unroll 1
  for (int h = 0; h < 4; ++h) {
    const int hs = 4 * ti + h;
    const int hs_n = hs + 1 < n_steps ? hs + 1 : hs;        // (the last step re-requests its own rows: unused)
    const int tile_n = (int)blockIdx.x + (hs_n >> 2) * (int)gridDim.x, h_n = hs_n & 3;
    // the projection's accumulators belong in the accumulation half of the register file: left to itself hipcc keeps these loop-carried
    // values in architectural registers (copying them to AGPR temporaries around every MFMA) and spills the attention's operands
#pragma unroll
    for (int nb = 0; nb < 16; ++nb)
#pragma unroll
      for (int t = 0; t < NG; ++t) asm volatile("" : "+a"(acc[nb][t]));

    stamp(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this step's q rows, k / v regions and ring slabs g, g + 1 have landed
    stamp(1);

    // ================= attention of head h on the wave's T tokens =================
    // (ordered so that the values vector instructions touch stay well below the 256 architectural registers: K planes 48, one query
    // group's Q planes 16 and score tiles 12 at a time, P planes 36; sched_barriers keep hipcc from merging the phases)
    u32x4 oh[NG][2], ol[NG][2];                             // o^T planes: [token group][k32 step of the head]
    {
      u32x4 kh[NG][2], kl[NG][2];
      float sq, sk;
      {
        f32x4 kr[4][NG];
#pragma unroll
        for (int t = 0; t < NG; ++t)
#pragma unroll
          for (int fb = 0; fb < 4; ++fb) kr[fb][t] = *reinterpret_cast<const f32x4*>(krd + 4 * t * AT_VG + 256 * fb);
        float mq = 0.f, mk = 0.f;
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
          for (int t = 0; t < NG; ++t) { mq = amax4(qr[fb][t], mq); mk = amax4(kr[fb][t], mk); }
        mq = wave_max(mq); mk = wave_max(mk);
        sq = pow2_scale(mq, 13); sk = pow2_scale(mk, 13);
#pragma unroll
        for (int t = 0; t < NG; ++t)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            u32x2 h0, l0, h1, l1;
            split4s(kr[2 * j][t], sk, h0, l0); split4s(kr[2 * j + 1][t], sk, h1, l1);
            kh[t][j] = cat2(h0, h1); kl[t][j] = cat2(l0, l1);
          }
      }
      // exp2 of log2(e)-scaled logits; S^T / 8 with the operand scales undone
      const float ssc = 0.125f * 1.4426950408889634f / (sq * sk);
      __builtin_amdgcn_sched_barrier(0);
      // the k region is free (its rows sit in registers as planes): the NEXT head step's k rows go there now, four pieces per query group,
      // spread over the softmax -- the kernel is bound by the rate HBM requests can be issued at (one 1 KB piece per wave and ~360 cycles is
      // what the memory system sustains); pieces issued in a burst, or only inside the projection's slabs, stall the wave at the issue
      kv_tile = tile_n; kv_h = h_n;

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
        planes_fence();
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
        __builtin_amdgcn_sched_barrier(0);
        dma_k_piece(4 * qg);
        __builtin_amdgcn_sched_barrier(0);
        float mx = -3.0e38f;
#pragma unroll
        for (int kg = 0; kg < NG; ++kg)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = (kmask[qg] >> (4 * kg + i)) & 1u ? st[kg][i] * ssc : -3.0e38f;
            st[kg][i] = v;
            mx = fmaxf(mx, v);
          }
        mx = gmax(mx);
        __builtin_amdgcn_sched_barrier(0);
        dma_k_piece(4 * qg + 1);
        __builtin_amdgcn_sched_barrier(0);
        float sum = 0.f;
#pragma unroll
        for (int kg = 0; kg < NG; ++kg)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float e = __builtin_amdgcn_exp2f(st[kg][i] - mx);          // (masked entries: exp2(-3e38) = 0)
            st[kg][i] = e;
            sum += e;
          }
        sum = gsum(sum);
        __builtin_amdgcn_sched_barrier(0);
        dma_k_piece(4 * qg + 2);
        __builtin_amdgcn_sched_barrier(0);
        const float inv = 8192.f * __builtin_amdgcn_rcpf(sum);               // (the sum's rcp: 1 ulp, a common factor of the column)
#pragma unroll
        for (int kg = 0; kg < NG; ++kg) { unsigned h0, h1, l0, l1; split4m(st[kg] * inv, h0, h1, l0, l1); ph[kg][qg] = u32x2{h0, h1}; pl[kg][qg] = u32x2{l0, l1}; }
        __builtin_amdgcn_sched_barrier(0);
        dma_k_piece(4 * qg + 3);
        __builtin_amdgcn_sched_barrier(0);
      }
      // the next head step's q rows -> registers (12 loads; this step's are planes by now)
      load_q(tile_n, h_n);
      __builtin_amdgcn_sched_barrier(0);

      stamp(2);
      // ---- O^T = V^T P^T: tile (fb, qg), contraction over the keys: key groups (0, 1) as one k32 step, a third group as a second one
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
      float lm[NG];                                         // 0 for tokens past M (their o must not reach the recorded maximum)
#pragma unroll
      for (int t = 0; t < NG; ++t) lm[t] = 1.f;
      if (!full) {
#pragma unroll
        for (int t = 0; t < NG; ++t) lm[t] = tok0 + 16 * t + c < a.M ? 1.f : 0.f;
      }
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
        planes_fence();
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

    // ================= output projection: the head's four slabs (output features [64 k, 64 k + 64)) =================
    // Round 6: its MFMAs are inline asm with the accumulator tied in the accumulation half, the file is compiled with -mllvm -amdgpu-mfma-vgpr-form, and
    // every other MFMA of the kernel (S^T, O^T: results that vector instructions consume at once) writes VGPRs directly -- see atl.hip `project`.
#define AT_ACC(ACC, WA, WB) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(WA), "v"(WB))
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      stamp(k4 == 0 ? 3 : 5);
      // slabs 2, 3 were requested during slabs 0, 1 of THIS step: at least 8 younger requests follow each of them (1 v piece behind the slab's
      // last ring piece, 4 ring + 3 v pieces of the next slab)
      if (k4 >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      __builtin_amdgcn_s_barrier();                         // slab gs is complete in LDS; every wave has left slab gs - 1
      stamp(4);
      ring_begin();                                         // slab gs + 2 goes into the slot of slab gs - 1 (past the last tile: bytes nobody reads)
      const char* sl = rd + (gs % AT_R) * AT_SLAB;
      // macro-step m = (nbl, j): fragments (hi, lo) of 16 output features x 32 k, 3 token groups x 3 products.  Order inside a macro-step,
      // pinned: first MFMA | this macro-step's LDS-DMA pieces | fragment reads of macro-step m + 1 | the other 8 MFMAs
      u32x4 wf[2][2];
      wf[0][0] = *reinterpret_cast<const u32x4*>(sl); wf[0][1] = *reinterpret_cast<const u32x4*>(sl + 1024);
      asm volatile("s_nop 1" ::: "memory");                   // (a just-written operand plane of o -> MFMA operand inside asm)
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int nb = 4 * k4 + (m >> 1), j = m & 1;
        const u32x4 wh = wf[m & 1][0], wl = wf[m & 1][1];
        __builtin_amdgcn_sched_barrier(0);
        AT_ACC(acc[nb][0], wh, ol[0][j]);
        __builtin_amdgcn_sched_barrier(0);
        if (!(m & 1) && !(ABL & 4)) ring_piece(m >> 1);                           // ring pieces at m = 0, 2, 4, 6
        if (!(ABL & 2) && (m == 1 || m == 4 || m == 7) && 3 * k4 + m / 3 < T / 4) dma_v_piece(3 * k4 + m / 3);      // v pieces 3 k4 + (0, 1, 2)
        __builtin_amdgcn_sched_barrier(0);
        if (m + 1 < 8) {
          wf[(m + 1) & 1][0] = *reinterpret_cast<const u32x4*>(sl + ((m + 1) * 2) * 1024);
          wf[(m + 1) & 1][1] = *reinterpret_cast<const u32x4*>(sl + ((m + 1) * 2 + 1) * 1024);
        }
        AT_ACC(acc[nb][0], wl, oh[0][j]);
        AT_ACC(acc[nb][0], wh, oh[0][j]);
#pragma unroll
        for (int t = 1; t < NG; ++t) {
          AT_ACC(acc[nb][t], wh, ol[t][j]);
          AT_ACC(acc[nb][t], wl, oh[t][j]);
          AT_ACC(acc[nb][t], wh, oh[t][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      ++gs;
    }

    stamp(5);
  }
    // ================= epilogue of the tile: y[tok][16 nb + 4 g ..] = acc * os + bias + constant of the row's variant + residual =================
    {
      asm volatile("s_nop 11" ::: "memory");                  // (the last asm MFMA's accumulator -> this block's reads: 12 wait states)
      __builtin_amdgcn_sched_barrier(0);
      const float* bsr = reinterpret_cast<const float*>(smem + AT_BIAS) + 4 * g;
      unsigned yoff[NG]; int rbo[NG];
#pragma unroll
      for (int t = 0; t < NG; ++t) {
        const unsigned tk = (unsigned)min((int)tok0 + 16 * t + c, m_last);      // (32-bit: a 64-bit division here costs hundreds of instructions and registers)
        yoff[t] = tk * 1024u + 16u * (unsigned)g;
        rbo[t] = RB ? a.rowvar[a.row0 + (int)(tk / (unsigned)a.L)] * 256 + 4 * g : 0;
      }
      const char* rbase = reinterpret_cast<const char*>(a.resid);
      char* ybase = reinterpret_cast<char*>(a.Y);
      if (full) {
        // all loads of a batch first, every store unconditional (a store behind a per-lane predicate sits in its own basic block behind
        // s_waitcnt vmcnt(0): DESIGN.md section 5); 8 batches of 2 feature blocks, the loads two batches ahead of their stores
        f32x4 rz[2][2][NG];
        auto rz_load = [&](int b2) __attribute__((always_inline)) {
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int t = 0; t < NG; ++t) rz[b2 & 1][q][t] = *reinterpret_cast<const f32x4*>(rbase + yoff[t] + 64 * (2 * b2 + q));
        };
        rz_load(0); rz_load(1);
#pragma unroll
        for (int b2 = 0; b2 < 8; ++b2) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int nb = 2 * b2 + q;
            const f32x4 bq = *reinterpret_cast<const f32x4*>(bsr + 16 * nb);
#pragma unroll
            for (int t = 0; t < NG; ++t) {
              f32x4 v = acc[nb][t] * os + bq + rz[b2 & 1][q][t];
              if (RB) v += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(smem + AT_RB) + rbo[t] + 16 * nb);
              *reinterpret_cast<f32x4*>(ybase + yoff[t] + 64 * nb) = v;
            }
          }
          if (b2 + 2 < 8) rz_load(b2 + 2);
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
          }
        }
      }
    }
  }
#undef AT_PIECE
#undef AT_ACC
  if (STAMP && a.stamps && lane == 0) {
    unsigned long long* o = a.stamps + ((long)blockIdx.x * 4 + wave) * 8;
    for (int k = 0; k < 6; ++k) o[k] = tk[k];
    o[6] = __builtin_amdgcn_s_memtime() - t_start;          // shader cycles of the whole kernel ..
    o[7] = __builtin_amdgcn_s_memrealtime() - r_start;      // .. over 100 MHz ticks: the clock it ran at
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no LDS-DMA may outlive the block

  amax = wave_max(amax);
  record_amax_block_guarded(a.amax_out, amax, reinterpret_cast<float*>(smem), a.range_flag, s_in, a.site);      // (no LDS-DMA in flight: vmcnt(0) above)
}

bool ato_applicable(int M, int L, int* ng) {
  int n = 0;
  if (L >= 1 && 48 % L == 0) n = 3; else if (L >= 1 && 32 % L == 0) n = 2;
  if (ng) *ng = n;
  // (the kernels address rows with 32-bit byte offsets into qkv, 3072 bytes per token: launches past that bound stay on the
  //  attention + tile / token-owning pair instead of failing)
  return n != 0 && M > 0 && M % L == 0 && (long)M * 3072 < (1l << 32);
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
  if (a.stamps) {
    RAMP_REQUIRE(ng == 3 && a.rowbias, "ato: the stamped twins exist for T = 48 with the row-variant constant");
    if (a.ablate == 2) hipLaunchKernelGGL((ato_kernel<3, true, 3>), dim3(nb), dim3(256), AT_LDS, s, a, n_tiles);
    else if (a.ablate == 4) hipLaunchKernelGGL((ato_kernel<3, true, 5>), dim3(nb), dim3(256), AT_LDS, s, a, n_tiles);
    else if (a.ablate == 6) hipLaunchKernelGGL((ato_kernel<3, true, 7>), dim3(nb), dim3(256), AT_LDS, s, a, n_tiles);
    else hipLaunchKernelGGL((ato_kernel<3, true, 1>), dim3(nb), dim3(256), AT_LDS, s, a, n_tiles);
  }
  else if (ng == 3) { if (a.rowbias) AT_GO(3, true); else AT_GO(3, false); }
  else { if (a.rowbias) AT_GO(2, true); else AT_GO(2, false); }
#undef AT_GO
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

int init_atk_attributes() {
#define AT_ATTR3(A) RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ato_kernel<3, true, A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AT_LDS))
#define AT_ATTR(NGV, RBV) RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ato_kernel<NGV, RBV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AT_LDS))
  AT_ATTR(3, true); AT_ATTR(3, false); AT_ATTR(2, true); AT_ATTR(2, false);
#undef AT_ATTR
  AT_ATTR3(1); AT_ATTR3(3); AT_ATTR3(5); AT_ATTR3(7);
  return 0;
}

}  // namespace ramp

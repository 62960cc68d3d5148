// Fused feed-forward of one transformer block, both directions, as TOKEN-OWNING waves (layers_attention_mini.py:38-45,
// 130-149: z2 = z1 + W2 (a * gelu(g)) + b2, [a | g] = W1 LN3(z1) + b1, and its input gradient).
//
// Why another dataflow (profiles/r02_*): the tile kernels move every K = 256 layer through HBM (1 KB in, 1-8 KB out per
// token and layer) and re-stage the activation tile once per 256 output columns; the two-role fused forward kernel
// (gemm.hip, ff_fwd_kernel) still round-trips the hidden through LDS, re-splits the activation tile eight times per M-tile
// and its two roles' work adds up.  Here nothing but the weights moves during the 32 hidden-unit steps of a token tile:
//
//   * a block = 4 waves = 128 tokens, wave w OWNS tokens [32 w, 32 w + 32) of the tile for the whole feed-forward;
//   * every product is computed TRANSPOSED, D^T[feature][token] = W[feature][k] * X^T[k][token]: the weight fragment is
//     the MFMA's A operand, the wave's tokens are the B operand, and the accumulator holds, per lane, ONE token and 16
//     features.  The LayerNorm output (forward) / dz (backward) of the wave's 32 tokens stays in registers for the whole
//     tile as two scaled fp16 planes (128 VGPRs), the 256-wide result accumulates in 128 more;
//   * GEGLU is lane-local: the a- and g-accumulators of a hidden unit hold the same (token, j) in the same lane and
//     register, and after the split into planes those registers ARE the B operand of the second product (the k order of
//     its weight fragments is permuted at pack time to the accumulator's row order) -- the hidden never leaves registers;
//   * the VJP stash [gelu(g) | a gelu'(g)] is written in the accumulator's own layout (1 KB per store instruction, fully
//     coalesced) and read back the same way by the backward kernel; no other kernel ever looks at it;
//   * LayerNorm-3 forward (prologue) and backward (epilogue) run on the registers the tile already holds: a token's row is
//     split over lanes r and r + 32, one shuffle per row sum.  ln_fwd / ln_bwd and their HBM round trips disappear;
//   * the weights (3 MB of fp16 planes per direction) stream through a 4-slot x 32 KB LDS ring filled by LDS-DMA
//     (global_load_lds, 1 KB per wave instruction, each wave issues a quarter of every slab): every fragment is fetched
//     ONCE per 128 tokens instead of once per 64, costs no VGPR and no VALU, and is read by ds_read_b128 at lane * 16.
//     One workgroup barrier per slab (= 48 MFMAs per wave), placed so that the next slab's first fragments are read
//     before it (no bubble): the barrier at the top of slab g certifies slab g + 1 and frees the slot of slab g - 1.
//
// Arithmetic is the fp16x3 split of gemm.hip (two scaled fp16 planes per operand, h1h1' + h1h2' + h2h1', fp32 accumulate),
// same delayed per-call-site scales, maxima recording and range guard: the kernels consume the call sites of the launches
// they replace (forward: FF1, FF2; backward: d(hg), FF1-dX) in the same order.
#include "args_token.h"
#include "pack.h"
#include "tokmma.h"

#include <algorithm>
#include <type_traits>

namespace ramp {

namespace {

constexpr int FX_SLAB = 32 * 1024;                      // bytes per ring slot: 8 macro-steps x 4 fragments x 1 KB
constexpr int FX_R = 4;                                 // ring slots
constexpr int FX_B1 = FX_R * FX_SLAB;                   // forward: the packed b1 (2048 floats) behind the ring
constexpr int FX_LN = FX_B1 + 2048 * 4;                   // LayerNorm-3 gamma (256) and beta (256) behind b1
constexpr int FX_B2 = FX_LN + 512 * 4;                     // forward: b2 (256)
constexpr size_t FX_LDS = (size_t)FX_R * FX_SLAB + 2048 * 4 + 512 * 4 + 256 * 4;
constexpr int FX_SLABS = 96;                            // slabs per 128-token tile (32 units x 3)
static_assert(FX_LDS <= 160 * 1024, "LDS budget");

}  // namespace

// slab kinds: which product a 32 KB slab of weight fragments feeds
enum { K_P1A = 0, K_P1B = 1, K_P2A = 2, K_P2B = 3 };
template <int V> using FO = std::integral_constant<int, V>;

// ABL (diagnostic twins for ramp_bench_gemm, wrong results; 0 in the product): 1 no LDS-DMA after the first slabs, 2 no stash
// traffic, 4 no elementwise step, 8 no slab barrier -- compile-time, a run-time test inside the slabs would split their
// scheduling regions
template <bool BWD, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void ffx_kernel(FfxArgs f, int n_mt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int n_my = (n_mt - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // >= 1 (grid <= n_mt)
  const int total_slabs = n_my * FX_SLABS;

  const float s_1 = scale_of(f.amax_in1), s_2 = scale_of(f.amax_in2);
  const float os1 = f.wsi1 / s_1, os2 = f.wsi2 / s_2;
  float amax1 = 0.f, amax2 = 0.f;
  const unsigned long long t_start = (ABL & 64) ? __builtin_amdgcn_s_memtime() : 0, r_start = (ABL & 64) ? __builtin_amdgcn_s_memrealtime() : 0;

  // ---- weight ring: the tile's 96 slabs lie in consumption order in ONE 3 MB stream (ffx_build_stream), so a slab is one
  // pointer step: wave w copies bytes [8 w KB, 8 w KB + 8 KB) of it as 8 LDS-DMA pieces of 1 KB, one per macro-step.
  // The pieces are inline asm: with __builtin_amdgcn_global_load_lds in the kernel hipcc stops counting LDS returns and waits
  // lgkmcnt(0) before every fragment use.  The compiler's own vmcnt waits stay safe: operations it does not know about are
  // younger or older IN ORDER, so its counts can only over-wait.  m0 = LDS address of the piece group; the instruction offset
  // applies to both addresses.  (Measured alternative: global_load_dwordx4 into staging registers + ds_write_b128 one slab
  // later -- two instructions the compiler can place freely -- is 15 % slower than the DMA.)
  const char* wsrc = reinterpret_cast<const char*>(f.Wstream) + wave * 8192 + lane * 16;
  int is_q = 0, is_g = 0;                                   // next slab to issue: position in the tile's sequence, global count
  const char* cur_src = wsrc; unsigned cur_dst = 0;        // the slab being issued piece by piece
  auto dma_begin = [&]() __attribute__((always_inline)) {
    cur_src = wsrc + (long)is_q * FX_SLAB;
    cur_dst = (unsigned)(uintptr_t)(smem + (is_g & (FX_R - 1)) * FX_SLAB + wave * 8192);
    is_q = is_q + 1 == FX_SLABS ? 0 : is_q + 1;
    ++is_g;
  };
#define FX_PIECE(C) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2" \
                                 :: "v"(cur_src + ((C) >> 2) * 4096), "s"(cur_dst + ((C) >> 2) * 4096), "n"(((C) & 3) * 1024) : "memory", "m0")
  auto dma_piece = [&](int c) __attribute__((always_inline)) {       // c: compile-time after unrolling
    switch (c) { case 0: FX_PIECE(0); break; case 1: FX_PIECE(1); break; case 2: FX_PIECE(2); break; case 3: FX_PIECE(3); break;
                 case 4: FX_PIECE(4); break; case 5: FX_PIECE(5); break; case 6: FX_PIECE(6); break; default: FX_PIECE(7); break; }
  };
  auto issue_slab = [&]() __attribute__((always_inline)) {           // (prologue only: a whole slab at once)
    dma_begin();
#pragma unroll
    for (int c = 0; c < 8; ++c) dma_piece(c);
  };

  // ---- consumer state ------------------------------------------------------------------------------------------------
  u32x4 F[3][4];                                            // fragment ring: macro-step m of a slab in F[(m + fo) % 3], two steps ahead of the MFMAs
  int g = 0;                                                // slabs consumed so far (ring position)
  const char* rd = smem + lane * 16;
  auto read_macro = [&](u32x4 (&dst)[4], int slot, int m) __attribute__((always_inline)) {
    const char* p = rd + slot * FX_SLAB + m * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = *reinterpret_cast<const u32x4*>(p + i * 1024);
  };
  // top of slab g: slab g + 1 is certified (every wave's share has landed: its pieces were issued during slab g - 2, the only
  // younger LDS-DMA are my 8 pieces of slab g + 2), the slot of slab g - 1 is free -> slab g + 3, one piece per macro-step of
  // slab g, behind an MFMA (issued back to back at the slab top the eight pieces held the wave for ~700 cycles).  The pieces
  // are issued unconditionally: past the block's last slab they refill the free slot with bytes nobody reads.
  unsigned long long tk[4] = {0, 0, 0, 0}, tlast = 0;
  auto slab_top = [&](auto vm_c) __attribute__((always_inline)) {
    constexpr int VM = decltype(vm_c)::value;               // 8; 16 where the eight stash loads of the backward sit between the pieces
    if (ABL & 64) {                                          // diagnostic: where a slab's cycles go (s_memtime perturbs the LDS waits a little)
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      if (tlast) tk[3] += t0 - tlast;
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VM) : "memory");
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();
      const unsigned long long t2 = __builtin_amdgcn_s_memtime();
      dma_begin();
      tlast = __builtin_amdgcn_s_memtime();
      tk[0] += t1 - t0; tk[1] += t2 - t1; tk[2] += tlast - t2;
      return;
    }
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VM) : "memory");
    if (!(ABL & 8)) __builtin_amdgcn_s_barrier();
    dma_begin();
  };

  u32x4 XB[16][2];                                          // the wave's tokens as B operand: [k16 step][plane]
  f32x16 acc2[8];                                           // 256 features x 32 tokens
  f32x16 acc1[2][BWD ? 1 : 2];                              // [unit parity][a, g] (backward: d(hg))
  u32x4 HB[BWD ? 4 : 2][2];                                 // the second product's B operand: [k16 step][plane]
  f32x4 st1[BWD ? 4 : 1], st2[BWD ? 4 : 1];                 // backward: the stash of the unit E works on next

  // six MFMAs of a macro-step: small terms first (lo x hi, hi x lo, hi x hi), two k16 steps or two accumulators.
  // Issue order, pinned with sched_barrier: MFMA 1 | two ds_read_b128 | MFMAs 2..4 | two ds_read_b128 | MFMAs 5, 6 + side work
  // (at most two fragment reads per MFMA gap: a third saturates the LDS array).
  const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define FX_MAC_HEAD(X, BX, FB, Z) X = mfma16(FB[1], BX[0], (Z) ? zero16 : X)
#define FX_MAC_MID(X, Y, BX, BY, FB, Z)                                \
  do {                                                                 \
    X = mfma16(FB[0], BX[1], X); X = mfma16(FB[0], BX[0], X);          \
    Y = mfma16(FB[3], BY[0], (Z) ? zero16 : Y);                        \
  } while (0)
#define FX_MAC_TAIL(Y, BY, FB) do { Y = mfma16(FB[2], BY[1], Y); Y = mfma16(FB[2], BY[0], Y); } while (0)

  // one slab = 8 macro-steps.  The fragment reads run TWO macro-steps ahead of the MFMAs (three buffers: the registers a
  // read overwrites were last used two steps ago, and its data has ~380 cycles to land): entering a slab, its macro 0 and 1
  // are already in F[FO % 3], F[(FO + 1) % 3]; leaving it, macro 0 and 1 of the NEXT slab are (8 = 2 mod 3: FO advances by 2).
  auto slab = [&](auto kind_c, auto fo_c, int par, auto side, auto vm_c) __attribute__((always_inline)) {
    constexpr int KIND = decltype(kind_c)::value, FO = decltype(fo_c)::value;
    slab_top(vm_c);
    const int slot = g & (FX_R - 1), nslot = (g + 1) & (FX_R - 1);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      u32x4 (&FB)[4] = F[(m + FO) % 3];
      u32x4 (&FN)[4] = F[(m + 2 + FO) % 3];
      const char* np = rd + (m < 6 ? slot * FX_SLAB + (m + 2) * 4096 : nslot * FX_SLAB + (m - 6) * 4096);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!BWD) {
        if constexpr (KIND == K_P1A || KIND == K_P1B) FX_MAC_HEAD(acc1[par][0], XB[8 * KIND + m], FB, KIND == K_P1A && m == 0);
        else FX_MAC_HEAD(acc2[m], HB[0], FB, false);
      } else {
        if constexpr (KIND == K_P1A) FX_MAC_HEAD(acc1[par][0], XB[2 * m], FB, m == 0);
        else FX_MAC_HEAD(acc2[4 * (KIND - K_P2A) + (m >> 1)], HB[2 * (m & 1)], FB, false);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!(ABL & 1)) dma_piece(m);
      __builtin_amdgcn_sched_barrier(0);
      // ---- one scheduling region: the four fragment reads, MFMAs 2..6 and this step's share of the elementwise work ----
#pragma unroll
      for (int i = 0; i < 4; ++i) FN[i] = *reinterpret_cast<const u32x4*>(np + i * 1024);
      if constexpr (!BWD) {
        if constexpr (KIND == K_P1A || KIND == K_P1B) {
          const int s = 8 * KIND + m;
          FX_MAC_MID(acc1[par][0], acc1[par][1], XB[s], XB[s], FB, KIND == K_P1A && m == 0); FX_MAC_TAIL(acc1[par][1], XB[s], FB);
        } else { FX_MAC_MID(acc2[m], acc2[m], HB[0], HB[1], FB, false); FX_MAC_TAIL(acc2[m], HB[1], FB); }
      } else {
        if constexpr (KIND == K_P1A) { FX_MAC_MID(acc1[par][0], acc1[par][0], XB[2 * m], XB[2 * m + 1], FB, false); FX_MAC_TAIL(acc1[par][0], XB[2 * m + 1], FB); }
        else {
          const int kb = 4 * (KIND - K_P2A) + (m >> 1), t = 2 * (m & 1);
          FX_MAC_MID(acc2[kb], acc2[kb], HB[t], HB[t + 1], FB, false); FX_MAC_TAIL(acc2[kb], HB[t + 1], FB);
        }
      }
      side(m);
      // issue order hints: 2 reads | MFMA + up to 7 vector instructions, five times, the other 2 reads after the second MFMA
      // (at most two fragment reads per MFMA gap: a third saturates the LDS array)
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 5, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x402, 5, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 5, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 5, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 12, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    ++g;
  };
  auto no_side = [](int) __attribute__((always_inline)) {};

  // ---- prologue: b1 into LDS, first three slabs in flight --------------------------------------------------------------
  if (!BWD) {
    float* b1s = reinterpret_cast<float*>(smem + FX_B1);
    for (int i = tid; i < 512; i += 256) reinterpret_cast<f32x4*>(b1s)[i] = reinterpret_cast<const f32x4*>(f.b1)[i];
    reinterpret_cast<float*>(smem + FX_B2)[tid] = f.b2[tid];
  }
  {   // gamma | beta in LDS for the backward epilogue.  (The forward prologue reads them from global memory: 64 L1-resident
      // loads per tile and lane that hipcc hoists above the row statistics; the same reads from LDS measured 2 % slower END TO
      // END -- ABL 16 is that variant.)
    float* lns = reinterpret_cast<float*>(smem + FX_LN);
    if (tid < 64) reinterpret_cast<f32x4*>(lns)[tid] = reinterpret_cast<const f32x4*>(f.ln_g)[tid];
    else if (tid < 128 && f.ln_b) reinterpret_cast<f32x4*>(lns)[tid] = reinterpret_cast<const f32x4*>(f.ln_b)[tid - 64];
  }
  issue_slab(); issue_slab(); issue_slab();
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");          // slab 0 (my share)
  __syncthreads();                                           // (also publishes b1 in LDS)
  read_macro(F[0], 0, 0); read_macro(F[1], 0, 1);

  const float* b1s = reinterpret_cast<const float*>(smem + FX_B1);
  const float* lng = reinterpret_cast<const float*>(smem + FX_LN);
  const float* lnb = lng + 256;

  for (int ti = 0; ti < n_my; ++ti) {
    const int mt = (int)blockIdx.x + ti * (int)gridDim.x;
    const long tok = (long)mt * 128 + wave * 32 + r;
    const bool tok_ok = tok < f.M;
    const long tokc = tok_ok ? tok : f.M - 1;
    float* stash_w = f.stash + (((long)mt * 32) * 4 + wave) * 2048 + lane * 4;      // + unit * 8192 + (2 q + which) * 256

    // ---- the wave's 32 tokens -> B-operand planes (forward: through LayerNorm-3) ---------------------------------------
    {
      const float* xrow = f.X + tokc * 256 + 8 * h;          // lane (r, h) holds k = 16 s + 8 h + i of token r
      f32x4 xv[32];
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        xv[2 * s] = *reinterpret_cast<const f32x4*>(xrow + 16 * s);
        xv[2 * s + 1] = *reinterpret_cast<const f32x4*>(xrow + 16 * s + 4);
      }
      if (!BWD) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) sum += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.f / 256.f);
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d = xv[i][e] - mean; ss += d * d; }
        ss += __shfl_xor(ss, 32);
        const float rstd = 1.f / sqrtf(ss * (1.f / 256.f) + 1e-5f);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          const int k = 16 * (i >> 1) + 8 * h + 4 * (i & 1);
          const f32x4 gm = *reinterpret_cast<const f32x4*>(((ABL & 16) ? lng : f.ln_g) + k), bt = *reinterpret_cast<const f32x4*>(((ABL & 16) ? lnb : f.ln_b) + k);
#pragma unroll
          for (int e = 0; e < 4; ++e) xv[i][e] = (xv[i][e] - mean) * rstd * gm[e] + bt[e];
        }
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        amax_pin(amax1, xv[2 * s][0], xv[2 * s][1]); amax_pin(amax1, xv[2 * s][2], xv[2 * s][3]);      // (pinned: see tokmma.h)
        amax_pin(amax1, xv[2 * s + 1][0], xv[2 * s + 1][1]); amax_pin(amax1, xv[2 * s + 1][2], xv[2 * s + 1][3]);
        split8(xv[2 * s] * s_1, xv[2 * s + 1] * s_1, XB[s][0], XB[s][1]);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc2[i][e] = 0.f;

    auto stash_load = [&](int u) __attribute__((always_inline)) {                           // backward: prefetch unit u's stash
      if (BWD && !(ABL & 2)) {
        const float* p = stash_w + (long)u * 8192;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          st1[BWD ? q : 0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (2 * q) * 256));      // (read once)
          st2[BWD ? q : 0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (2 * q + 1) * 256));
        }
      }
    };
    // E(u): the elementwise step between the two products, on acc1[par], cut into work items that the slabs of a group
    // take one at a time (E_step): forward 16 elements (one GEGLU each; a quad's stash rows leave with its last element)
    // and 2 plane packs, backward 4 quads and 4 packs.
    f32x4 hq[BWD ? 8 : 4];                                   // forward: h quads; backward: [da quads | dg quads]
    f32x4 ba[2], bg[2];                                      // forward: b1 of quad q in [q & 1], read a stage ahead
    // forward: GEGLU of one quad (four independent dependency chains: a lone wave has nothing else to fill the vector
    // latencies with) in three stages of ~30 vector instructions, one stage per macro-step
    // (the bias quad of the NEXT stage 0 is read during stage 1: read where it is used, the fma behind it exposes the LDS latency)
    f32x4 qa[2], qg[2], qt[2], qp[2], s1q, s2q;              // the quad in progress [q & 1]: a, g, t = 1 / (1 + p |g|) then Phi(g), phi(g)
    auto bias_load = [&](int u, int q) __attribute__((always_inline)) {
      if constexpr (!BWD) {
        ba[q & 1] = *reinterpret_cast<const f32x4*>(b1s + (2 * u) * 32 + 8 * q + 4 * h);
        bg[q & 1] = *reinterpret_cast<const f32x4*>(b1s + (2 * u + 1) * 32 + 8 * q + 4 * h);
      }
    };
    bias_load(0, 0);
    auto geglu_stage = [&](int u, int par, int q, int st, int hf) __attribute__((always_inline)) {      // elements 2 hf, 2 hf + 1 of quad q
      if constexpr (!BWD) {
        if (ABL & 128) {                                     // diagnostic: two instructions per element instead of ~30
          if (st == 0) {
#pragma unroll
            for (int e = 2 * hf; e < 2 * hf + 2; ++e) hq[q][e] = fmaf(acc1[par][0][4 * q + e], os1, acc1[par][1][4 * q + e]) * s_2;
          }
          return;
        }
        if (st == 0) {
#pragma unroll
          for (int e = 2 * hf; e < 2 * hf + 2; ++e) {
            qa[q & 1][e] = fmaf(acc1[par][0][4 * q + e], os1, ba[q & 1][e]);
            qg[q & 1][e] = fmaf(acc1[par][1][4 * q + e], os1, bg[q & 1][e]);
            qt[q & 1][e] = __builtin_amdgcn_rcpf(fmaf(0.2316419f, fabsf(qg[q & 1][e]), 1.f));
            qp[q & 1][e] = __builtin_amdgcn_exp2f(qg[q & 1][e] * qg[q & 1][e] * -0.72134752044448170368f) * 0.39894228040143267794f;
          }
        } else if (st == 1) {
#pragma unroll
          for (int e = 2 * hf; e < 2 * hf + 2; ++e) {
            const float t = qt[q & 1][e];
            float poly = fmaf(1.330274429f, t, -1.821255978f);
            poly = fmaf(poly, t, 1.781477937f);
            poly = fmaf(poly, t, -0.356563782f);
            poly = fmaf(poly, t, 0.319381530f);
            const float qq = qp[q & 1][e] * (poly * t);
            qt[q & 1][e] = qg[q & 1][e] >= 0.f ? 1.f - qq : qq;      // Phi(g)
          }
          if (hf == 1) { if (q < 3) bias_load(u, q + 1); else bias_load(u < 31 ? u + 1 : 31, 0); }
        } else {
#pragma unroll
          for (int e = 2 * hf; e < 2 * hf + 2; ++e) {
            s1q[e] = qg[q & 1][e] * qt[q & 1][e];                            // gelu(g)
            s2q[e] = qa[q & 1][e] * fmaf(qg[q & 1][e], qp[q & 1][e], qt[q & 1][e]);   // a * gelu'(g)
            const float hv = qa[q & 1][e] * s1q[e];                          // a * gelu(g)
            hq[q][e] = hv;
          }
          amax_pin(amax2, hq[q][2 * hf], hq[q][2 * hf + 1]);
          hq[q][2 * hf] *= s_2; hq[q][2 * hf + 1] *= s_2;
          if (hf == 1 && !(ABL & 2)) {
            float* p = stash_w + (long)u * 8192;
            // non-temporal, like the backward's loads of it: the stash is written once and read once a whole pass later; keeping it
            // out of the L2's way measured +1.2 % (stores) and +1.3 % (loads, which are prefetched three slabs ahead) END TO END.
            // (The same hint on the kernels' outputs or on their exposed epilogue loads measured -1.2 ... -3.3 %: those are consumed
            // within microseconds / sit on the critical path.)
            __builtin_nontemporal_store(s1q, reinterpret_cast<f32x4*>(p + (2 * q) * 256));
            __builtin_nontemporal_store(s2q, reinterpret_cast<f32x4*>(p + (2 * q + 1) * 256));
          }
        }
      }
    };
    auto quad_b = [&](int par, int q) __attribute__((always_inline)) {
      if constexpr (BWD) {
        const f32x4 d = quad(acc1[par][0], q) * os1;
        const f32x4 da = d * st1[q], dg = d * st2[q];
        amax_pin(amax2, da[0], da[1]); amax_pin(amax2, da[2], da[3]); amax_pin(amax2, dg[0], dg[1]); amax_pin(amax2, dg[2], dg[3]);
        hq[q] = da * s_2; hq[4 + q] = dg * s_2;
      }
    };
    auto pack1 = [&](int t) __attribute__((always_inline)) { split8(hq[2 * t], hq[2 * t + 1], HB[t][0], HB[t][1]); };
    auto pack_half = [&](int t, int hf) __attribute__((always_inline)) {
      unsigned h0, h1, l0, l1;
      split4(hq[2 * t + hf], h0, h1, l0, l1);
      HB[t][0][2 * hf] = h0; HB[t][0][2 * hf + 1] = h1; HB[t][1][2 * hf] = l0; HB[t][1][2 * hf + 1] = l1;
    };
    // step idx of n: forward items = the quads' stages and the two packs (a pack not before step `pack_from`: the second
    // product still reads HB),
    // backward items q0..q3 [stash prefetch of unit u + 1] P0..P3, the packs not before step `pack_from`
    auto E_step = [&](int u, int par, int idx, int n, int pack_from) __attribute__((always_inline)) {
      if (ABL & 4) return;
      if constexpr (!BWD) {
#pragma unroll
        for (int it = 0; it < 28; ++it) {                    // q0: s0 h0, s0 h1, s1 h0, .. s2 h1; q1: ..; P0 h0, P0 h1; q2; q3; P1 h0, P1 h1
          const bool is_pack = (it >= 12 && it < 14) || it >= 26;
          int at = it * n / 28;
          if (is_pack && at < pack_from) at = pack_from;
          if (at > n - 1) at = n - 1;
          if (at != idx) continue;
          if (is_pack) pack_half(it >= 26 ? 1 : 0, it & 1);
          else { const int j = it < 12 ? it : it - 2; geglu_stage(u, par, j / 6, (j % 6) >> 1, j & 1); }
        }
      } else {
#pragma unroll
        for (int it = 0; it < 9; ++it) {
          int at = it < 4 ? it * pack_from / 4 : (it == 4 ? pack_from - 1 : pack_from + (it - 5) * (n - pack_from) / 4);
          if (at > n - 1) at = n - 1;
          if (at != idx) continue;
          if (it < 4) quad_b(par, it);
          else if (it == 4) { if (u + 1 < 32) stash_load(u + 1); }
          else pack1(it - 5);
        }
      }
    };

    // ---- the 32 hidden units, software-pipelined: P2(k - 1), P1(k + 1) and E(k) share a group ---------------------------
    using KA = std::integral_constant<int, K_P1A>; using KB = std::integral_constant<int, K_P1B>;
    using KC = std::integral_constant<int, K_P2A>; using KD = std::integral_constant<int, K_P2B>;
    using V8 = std::integral_constant<int, 8>; using V16 = std::integral_constant<int, 16>;
    // FO<x>: which fragment buffer holds the slab's macro-step 0 (every slab advances the ring by 8 = 2 mod 3)
    if constexpr (!BWD) {
      slab(KA{}, FO<0>{}, 0, no_side, V8{}); slab(KB{}, FO<2>{}, 0, no_side, V8{});
      slab(KA{}, FO<1>{}, 1, [&](int m) __attribute__((always_inline)) { E_step(0, 0, m, 16, 0); }, V8{});
      slab(KB{}, FO<0>{}, 1, [&](int m) __attribute__((always_inline)) { E_step(0, 0, 8 + m, 16, 0); }, V8{});
#pragma unroll 1
      for (int kk = 1; kk <= 29; kk += 2) {
#pragma unroll
        for (int o = 0; o < 2; ++o) {                        // k = kk + o: acc1 parity of E(k) is k & 1 = 1 - o
          const int k = kk + o, pe = 1 - o, pn = o;
          slab(KC{}, FO<2>{}, 0, [&](int m) __attribute__((always_inline)) { E_step(k, pe, m, 24, 8); }, V8{});
          slab(KA{}, FO<1>{}, pn, [&](int m) __attribute__((always_inline)) { E_step(k, pe, 8 + m, 24, 8); }, V8{});
          slab(KB{}, FO<0>{}, pn, [&](int m) __attribute__((always_inline)) { E_step(k, pe, 16 + m, 24, 8); }, V8{});
        }
      }
      slab(KC{}, FO<2>{}, 0, [&](int m) __attribute__((always_inline)) { E_step(31, 1, m, 9, 8); }, V8{});      // P2(30) with E(31)
      E_step(31, 1, 8, 9, 8);                                                                    // its packs, after P2(30)
      slab(KC{}, FO<1>{}, 0, no_side, V8{});                                                                    // P2(31)
    } else {
      stash_load(0);
      slab(KA{}, FO<0>{}, 0, no_side, V8{});
      slab(KA{}, FO<2>{}, 1, [&](int m) __attribute__((always_inline)) { E_step(0, 0, m, 8, 4); }, V8{});
#pragma unroll 1
      for (int kk = 1; kk <= 29; kk += 2) {
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const int k = kk + o, pe = 1 - o, pn = o;
          slab(KC{}, FO<1>{}, 0, [&](int m) __attribute__((always_inline)) { E_step(k, pe, m, 24, 16); }, V8{});
          slab(KD{}, FO<0>{}, 0, [&](int m) __attribute__((always_inline)) { E_step(k, pe, 8 + m, 24, 16); }, V8{});
          slab(KA{}, FO<2>{}, pn, [&](int m) __attribute__((always_inline)) { E_step(k, pe, 16 + m, 24, 16); }, V16{});   // (8 stash loads + 8 pieces younger)
        }
      }
      slab(KC{}, FO<1>{}, 0, [&](int m) __attribute__((always_inline)) { E_step(31, 1, m, 20, 16); }, V8{});
      slab(KD{}, FO<0>{}, 0, [&](int m) __attribute__((always_inline)) { E_step(31, 1, 8 + m, 20, 16); }, V8{});
#pragma unroll
      for (int i = 16; i < 20; ++i) E_step(31, 1, i, 20, 16);
      slab(KC{}, FO<2>{}, 0, no_side, V8{}); slab(KD{}, FO<1>{}, 0, no_side, V8{});
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

    // ---- epilogue: lane (r, h) holds features n = 32 nb + 8 q + 4 h + i of token r --------------------------------------
    // (the row addresses are recomputed from the tile index here -- opaque to hipcc, which otherwise carries them across the
    // 96 slabs in spilled registers and reloads them one vmcnt(0) at a time)
    int mt_e = mt;
    asm volatile("" : "+s"(mt_e));
    long tok_e = (long)mt_e * 128 + wave * 32 + r;
    tok_e = tok_e < f.M ? tok_e : f.M - 1;
    if (!BWD) {
      const float* zrow = f.Z1 + tok_e * 256 + 4 * h;
      float* orow = f.Y + tok_e * 256 + 4 * h;
      const float* b2s = reinterpret_cast<const float*>(smem + FX_B2) + 4 * h;
      // all 32 residual quads first, every store unconditional (tokens past M recompute and rewrite row M - 1 with the same
      // bits): with `if (tok_ok)` around each store hipcc put every quad's loads in their own basic block behind a vmcnt(0)
      // -- eight serial memory round trips per tile here, thirty-two in the backward epilogue
      f32x4 rz[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) rz[i] = *reinterpret_cast<const f32x4*>(zrow + 8 * i);
#pragma unroll
      for (int i = 0; i < 32; ++i) {                          // i = 4 nb + q
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(b2s + 8 * i);      // (LDS: a global load here would be waited for with vmcnt(0), one round trip per quad)
        const f32x4 v = quad(acc2[i >> 2], i & 3) * os2 + b2 + rz[i];
        *reinterpret_cast<f32x4*>(orow + 8 * i) = v;
      }
    } else {
      // dz1 = dz + LNbwd(d(ln3); z1, gamma)   (rowops.hip, ln_bwd_kernel)
      const float* zrow = f.Z1 + tok_e * 256 + 4 * h;
      const float* drow = f.X + tok_e * 256 + 4 * h;
      float* orow = f.Y + tok_e * 256 + 4 * h;
      f32x4 xz[32], add[32];
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 32; ++i) xz[i] = *reinterpret_cast<const f32x4*>(zrow + 8 * i);
#pragma unroll
      for (int i = 0; i < 32; ++i) add[i] = *reinterpret_cast<const f32x4*>(drow + 8 * i);
#pragma unroll
      for (int i = 0; i < 32; ++i) sum += (xz[i][0] + xz[i][1]) + (xz[i][2] + xz[i][3]);
      sum += __shfl_xor(sum, 32);
      const float mean = sum * (1.f / 256.f);
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 32; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = xz[i][e] - mean; ss += d * d; }
      ss += __shfl_xor(ss, 32);
      const float rstd = 1.f / sqrtf(ss * (1.f / 256.f) + 1e-5f);
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int i = 0; i < 32; ++i) {                          // i = 4 nb + q
        const f32x4 gm = *reinterpret_cast<const f32x4*>(lng + 8 * i + 4 * h);
        f32x4 gq = quad(acc2[i >> 2], i & 3) * os2 * gm;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xz[i][e] = (xz[i][e] - mean) * rstd;
          t1 += gq[e]; t2 += gq[e] * xz[i][e];
          acc2[i >> 2][4 * (i & 3) + e] = gq[e];
        }
      }
      t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
      const float m1 = t1 * (1.f / 256.f), m2 = t2 * (1.f / 256.f);
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (acc2[i >> 2][4 * (i & 3) + e] - m1 - xz[i][e] * m2) * rstd + add[i][e];
        *reinterpret_cast<f32x4*>(orow + 8 * i) = o;            // (unconditional: see the forward epilogue)
      }
    }
    // Retire all but the youngest 24 vector-memory operations before the next tile queues its row loads behind this tile's 32
    // stores and up to 16 LDS-DMA pieces (vmcnt is a 6-bit counter; round 2's parked LayerNorm-in-the-loader GEGLU variant had
    // 70 operations in flight at the point where its sporadic stale rows appeared -- a suspicion, never proven: issue should
    // simply stall at 63).  Free in the A/B (ABL 32 = without it): the stores it waits for are a tile old.
    if (!(ABL & 32)) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  }
#undef FX_PIECE
#undef FX_MAC_HEAD
#undef FX_MAC_MID
#undef FX_MAC_TAIL
  if ((ABL & 64) && f.stamps && lane == 0) {
    unsigned long long* o = f.stamps + ((long)blockIdx.x * 4 + wave) * 6;
    o[0] = tk[0]; o[1] = tk[1]; o[2] = tk[2]; o[3] = tk[3];
    o[4] = __builtin_amdgcn_s_memtime() - t_start;          // shader cycles of the whole kernel ..
    o[5] = __builtin_amdgcn_s_memrealtime() - r_start;      // .. over 100 MHz ticks: the clock it ran at
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no LDS-DMA may outlive the block

  // ---- maxima for the next evaluation's scales, range guard (as in gemm_x6p_body.inc) ---------------------------------
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { amax1 = fmaxf(amax1, __shfl_xor(amax1, o)); amax2 = fmaxf(amax2, __shfl_xor(amax2, o)); }
  record_amax_block(f.amax_out1, amax1, reinterpret_cast<float*>(smem));      // (no LDS-DMA in flight: vmcnt(0) above)
  record_amax_block(f.amax_out2, amax2, reinterpret_cast<float*>(smem) + 4);
  if (lane == 0) {
    if (f.range_flag) {
      if (!(amax1 * s_1 < 60000.f) || (amax1 > 0.f && amax1 * s_1 < 0.125f)) atomicMax(f.range_flag, f.site1 + 1);
      if (!(amax2 * s_2 < 60000.f) || (amax2 > 0.f && amax2 * s_2 < 0.125f)) atomicMax(f.range_flag, f.site2 + 1);
    }
  }
}

// out[row][c] = in[row][src(c)]: the k order of the second product's weight fragments follows the accumulator rows of
// the first (element i of lane half h in k16 step s of a 32-block <-> row 4 h + 8 (2 s + (i >> 2)) + (i & 3)).
//   mode 0 (forward,  W2  [256][1024]): blocks of 32 columns, 2 steps each
//   mode 1 (backward, W1^T [256][2048]): per unit u 64 columns = steps 0, 1 from a-columns 32 u.., steps 2, 3 from g-columns 1024 + 32 u..
__global__ void ffx_gather_cols_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols, int mode) {
  const long total = (long)rows * cols;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long row = idx / cols; const int c = (int)(idx - row * cols);
    const int i = c & 7, hh = (c >> 3) & 1;
    int src;
    if (mode == 0) { const int u = c >> 5, s = (c >> 4) & 1; src = 32 * u + 4 * hh + 8 * (2 * s + (i >> 2)) + (i & 3); }
    else { const int u = c >> 6, t = (c >> 4) & 3; src = (t >= 2 ? 1024 : 0) + 32 * u + 4 * hh + 8 * (2 * (t & 1) + (i >> 2)) + (i & 3); }
    out[idx] = in[row * cols + src];
  }
}
int ffx_pack_second(const float* W, int rows, int cols, int mode, float scale, float* tmp, unsigned short* out, hipStream_t s) {
  RAMP_REQUIRE(W && tmp && out && rows % 32 == 0 && ((mode == 0 && cols % 32 == 0) || (mode == 1 && cols == 2048)), "ffx_pack_second: bad shape");
  hipLaunchKernelGGL(ffx_gather_cols_kernel, dim3(1024), dim3(256), 0, s, W, tmp, rows, cols, mode);
  RAMP_HIP_CHECK(hipGetLastError());
  return launch_pack_h3(tmp, out, rows, cols, scale, s);
}

// The weight stream of one direction: 96 slabs x 32 KB in the order the kernel consumes them; a slab = 8 macro-steps x
// 4 fragments (1 KB each, [lane][8 halves] as launch_pack_h3 writes them).
//   forward : P1(0)a P1(0)b | P1(1)a P1(1)b | { P2(k-1) P1(k+1)a P1(k+1)b } k = 1..30 | P2(30) | P2(31)
//             P1(u)x: macro m = k16 step 8 x + m of W1's row blocks 2 u (a) and 2 u + 1 (g): [a hi, a lo, g hi, g lo]
//             P2(u) : macro m = row block m of W2 (permuted), k16 steps 2 u, 2 u + 1: [s0 hi, s0 lo, s1 hi, s1 lo]
//   backward: P1(0) | P1(1) | { P2(k-1)a P2(k-1)b P1(k+1) } k = 1..30 | P2(30)a P2(30)b | P2(31)a P2(31)b
//             P1(u) : macro m = k16 steps 2 m, 2 m + 1 of W2^T's row block u
//             P2(u)x: macro m = row block 4 x + (m >> 1) of W1^T (permuted), k16 steps 4 u + 2 (m & 1), + 1
__global__ void ffx_build_stream_kernel(const unsigned short* __restrict__ p1, const unsigned short* __restrict__ p2,
                                        unsigned short* __restrict__ out, int bwd) {
  const int q = blockIdx.x;                                  // slab
  int kind, u;
  if (!bwd) {
    if (q < 4) { kind = q & 1; u = q >> 1; }
    else if (q < 94) { const int k = 1 + (q - 4) / 3, j = (q - 4) - 3 * (k - 1); if (j == 0) { kind = K_P2A; u = k - 1; } else { kind = j - 1; u = k + 1; } }
    else { kind = K_P2A; u = q - 64; }
  } else {
    if (q < 2) { kind = K_P1A; u = q; }
    else if (q < 92) { const int k = 1 + (q - 2) / 3, j = (q - 2) - 3 * (k - 1); if (j == 2) { kind = K_P1A; u = k + 1; } else { kind = K_P2A + j; u = k - 1; } }
    else { kind = K_P2A + (q & 1); u = 30 + ((q - 92) >> 1); }
  }
  for (int i = threadIdx.x >> 6; i < 32; i += blockDim.x >> 6) {       // fragment i = 4 m + j of the slab
    const int m = i >> 2, j = i & 3, lane = threadIdx.x & 63;
    const unsigned short* src;
    if (!bwd) {
      if (kind <= K_P1B) src = p1 + ((long)(((2 * u + (j >> 1)) * 16 + 8 * kind + m) * 2 + (j & 1))) * 512;
      else src = p2 + ((long)((m * 64 + 2 * u + (j >> 1)) * 2 + (j & 1))) * 512;
    } else {
      if (kind == K_P1A) src = p1 + ((long)((u * 16 + 2 * m + (j >> 1)) * 2 + (j & 1))) * 512;
      else src = p2 + ((long)(((4 * (kind - K_P2A) + (m >> 1)) * 128 + 4 * u + 2 * (m & 1) + (j >> 1)) * 2 + (j & 1))) * 512;
    }
    reinterpret_cast<u32x4*>(out + ((long)q * 32 + i) * 512)[lane] = reinterpret_cast<const u32x4*>(src)[lane];
  }
}
int ffx_build_stream(const unsigned short* p1, const unsigned short* p2, unsigned short* out, bool bwd, hipStream_t s) {
  RAMP_REQUIRE(p1 && p2 && out, "ffx_build_stream: null operand");
  hipLaunchKernelGGL(ffx_build_stream_kernel, dim3(FX_SLABS), dim3(256), 0, s, p1, p2, out, bwd ? 1 : 0);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_ffx(const FfxArgs& f, bool bwd, hipStream_t s) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  RAMP_REQUIRE(f.M > 0 && f.X && f.Y && f.Z1 && f.stash && f.Wstream && f.ln_g && (bwd || (f.ln_b && f.b1 && f.b2)), "ffx: null operand");
  RAMP_REQUIRE(al16(f.X) && al16(f.Y) && al16(f.Z1) && al16(f.stash) && al16(f.Wstream) && al16(f.ln_g) && al16(f.ln_b) &&
               al16(f.b1) && al16(f.b2), "ffx: operands must be 16-byte aligned");
  {   // rows past M are recomputed and rewritten from the inputs (unconditional stores): the output may alias none of them
    const size_t yb = (size_t)f.M * 256 * 4;
    RAMP_REQUIRE(!ranges_overlap(f.Y, yb, f.X, yb) && !ranges_overlap(f.Y, yb, f.Z1, yb), "ffx: the output must not overlap X or z1 (no in-place use)");
  }
  const int n_mt = (f.M + 127) / 128;
  const int nb = std::min(n_mt, device_cu_count());          // one 4-wave block per CU
#define FX_GO(B, A) hipLaunchKernelGGL((ffx_kernel<B, A>), dim3(nb), dim3(256), FX_LDS, s, f, n_mt)
  if (f.ablate == 0) { if (bwd) FX_GO(true, 0); else FX_GO(false, 0); }
  else if (f.ablate == 1) { if (bwd) FX_GO(true, 1); else FX_GO(false, 1); }
  else if (f.ablate == 2) { if (bwd) FX_GO(true, 2); else FX_GO(false, 2); }
  else if (f.ablate == 64) { if (bwd) FX_GO(true, 64); else FX_GO(false, 64); }
  else if (f.ablate == 66) { if (bwd) FX_GO(true, 66); else FX_GO(false, 66); }
  else if (f.ablate == 16) { if (bwd) FX_GO(true, 16); else FX_GO(false, 16); }
  else if (f.ablate == 32) { if (bwd) FX_GO(true, 32); else FX_GO(false, 32); }
  else if (f.ablate == 48) { if (bwd) FX_GO(true, 48); else FX_GO(false, 48); }
  else RAMP_REQUIRE(false, "ffx: ablation variant not built");
#undef FX_GO
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

int init_ffx_attributes() {
#define FX_ATTR(B, A) RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffx_kernel<B, A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FX_LDS))
  FX_ATTR(false, 0); FX_ATTR(true, 0); FX_ATTR(false, 1); FX_ATTR(true, 1); FX_ATTR(false, 2); FX_ATTR(true, 2); FX_ATTR(false, 64); FX_ATTR(true, 64); FX_ATTR(false, 66); FX_ATTR(true, 66); FX_ATTR(false, 16); FX_ATTR(true, 16); FX_ATTR(false, 32); FX_ATTR(true, 32); FX_ATTR(false, 48); FX_ATTR(true, 48);
#undef FX_ATTR
  return 0;
}

}  // namespace ramp

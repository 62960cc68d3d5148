// The token-owning fused feed-forward of ffx.hip on v_mfma_f32_16x16x32_f16 (round 6).  Same algorithm, same dataflow, same
// call sites, same stash size -- layers_attention_mini.py:38-45, 130-149: z2 = z1 + W2 (a * gelu(g)) + b2, [a | g] = W1 LN3(z1)
// + b1, and its input gradient -- re-tiled for the 16 x 16 x 32 instruction:
//
// Why: the kernel pair is half of a sampling job and runs power-limited (1.5 GHz in-kernel under the socket cap).  The MFMA +
// fragment-read loop of this dataflow holds a 14 % higher clock in the 16 x 16 x 32 shape at equal cycles per FLOP
// (ramp_amd/tools/mfma_shape_probe.hip, profiles/r06_mfma_shape_probe.txt: 1.75 against 1.53 GHz bare, +12 % with the ring
// refill, +10 % / +7 % with the backward's / forward's vector work beside it; MI355X_MICROARCH.md, DVFS give-back (7)).
//
// What changes against ffx.hip (read its header first):
//   * a weight fragment (one ds_read_b128, 1 KB) is 16 features x 32 k instead of 32 x 16 and feeds TWO MFMAs per product
//     term -- the wave's token halves 0-15 / 16-31 -- so LDS bytes per FLOP, the B-operand registers (32 tokens x 256 k as two
//     planes = 128) and the accumulator count stay what they were; a macro-step is 4 fragments and 12 MFMAs (was 6);
//   * lane (c, gq) = (lane & 15, lane >> 4) holds, of an operand, token c of a half and k = 8 gq .. 8 gq + 7 of a k32 step; of a
//     16 x 16 accumulator tile, token c and features 4 gq .. 4 gq + 3.  A token's 256-wide row is spread over FOUR lanes
//     (two permlane swaps per row sum) and every lane carries two tokens (one per half);
//   * a hidden unit (32 features) = feature tiles ft = 0, 1; its accumulator "quads" q = 2 t + ft (t = token half) are what
//     ffx.hip's quads 0..3 were: GEGLU stays lane-local, the quads 2 t, 2 t + 1 ARE the B operand (8 k values per lane) of the
//     second product's k32 step for token half t -- its weight fragments' k order is permuted at pack time to
//     k = 8 gq + jj  <->  hidden 16 (jj >> 2) + 4 gq + (jj & 3)  (ffx16_pack);
//   * the stash keeps its size and addressing ([tile][unit][wave][2 q + which][lane][4]); its element order inside a unit
//     follows the new accumulator layout -- private to this kernel pair as before (a forward launch of one shape and a
//     backward launch of the other do not mix: the engine picks one shape per context);
//   * the 4-register accumulators are pinned to the accumulation half of the file at the top of every loop trip
//     (asm "+a"): hipcc leaves the 16-register accumulators of ffx.hip in place by itself (tied operands) but moves
//     4-register ones through VGPRs at every back edge.
#include "args_token.h"
#include "pack.h"
#include "tokmma.h"
#include "atkmma.h"

#include <algorithm>
#include <type_traits>

namespace ramp {

namespace {

constexpr int F6_SLAB = 32 * 1024;                      // bytes per ring slot: 8 macro-steps x 4 fragments x 1 KB
constexpr int F6_R = 4;                                 // ring slots
constexpr int F6_B1 = F6_R * F6_SLAB;                   // forward: the packed b1 (2048 floats) behind the ring
constexpr int F6_LN = F6_B1 + 2048 * 4;                 // LayerNorm-3 gamma (256) and beta (256) behind b1
constexpr int F6_B2 = F6_LN + 512 * 4;                  // forward: b2 (256)
constexpr size_t F6_LDS = (size_t)F6_R * F6_SLAB + 2048 * 4 + 512 * 4 + 256 * 4;
constexpr int F6_SLABS = 96;                            // slabs per 128-token tile (32 units x 3)
static_assert(F6_LDS <= 160 * 1024, "LDS budget");

enum { Q_P1A = 0, Q_P1B = 1, Q_P2A = 2, Q_P2B = 3 };    // slab kinds (as ffx.hip)
template <int V> using QO = std::integral_constant<int, V>;

}  // namespace

namespace {
// two floats x s (s a power of two) -> one dword of each fp16 plane, four instructions: hi = rn16(x s) and lo = rn16(x s - hi) both by
// v_fma_mix{lo,hi}_f16 (fp32 fma rounded to fp16 once; x s and x s - hi are exact in fp32, so these are the bits of split4 on x s) -- no
// separate scale multiply, no conversions back (7 instructions per pair the plain way).  MEASURED, NOT USED: 3 % (forward) / 7 % (backward) fewer
// vector instructions in the loop and 0.4-0.9 % MORE time -- the asm pairs are invisible to sched_group_barrier and bunch where the compiler
// drops them, as round 4 found for the 32x32x16 kernels; kept as the ABL 1 twin
__device__ __forceinline__ void split2x(float x0, float x1, float s, unsigned& hi, unsigned& lo) {
  unsigned h, l;
  asm("v_fma_mixlo_f16 %0, %1, %3, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %2, %3, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]"
      : "=&v"(h) : "v"(x0), "v"(x1), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(l) : "v"(x0), "v"(x1), "v"(s), "v"(h));
  hi = h; lo = l;
}
}  // namespace

// ABL: 0 the product; 1 = the packs through split2x (A/B twin, same bits: measured 0.4-0.9 % SLOWER, profiles/r06_ab_same_box.txt); 64 = whole-kernel clock stamp per wave (s_memtime / s_memrealtime) into f.stamps [block][wave][6] slots 4, 5
// NT: token halves of a wave.  2 = the kernel described above (a wave owns 32 tokens, a tile is 128).  1 = HALF TILES (ffx16h_kernel): a wave owns 16
// tokens, a tile is 64 -- the same slabs with the second token half's MFMAs and elementwise work left out, i.e. more than half of a full tile's time
// (the fragment reads no longer amortise over two MFMAs), but on twice as many CUs: launch_ffx16 turns the tiles of a last round that would leave
// half of the CUs idle (the L = 6 level: 384 tiles on 256 CUs) into half tiles, so that round costs ~0.7 instead of 1 tile time.  n_mt counts the
// launch's tiles of ITS kind; f.tok0 / f.slot0 (NT = 1) are its first token and its first stash slot (two half tiles share a 128-token slot).
template <bool BWD, int ABL, int NT>
__device__ __forceinline__ void ffx16_body(const FfxArgs& f, int n_mt) {
  static_assert(NT == 1 || NT == 2, "token halves");
  constexpr int NQ = 2 * NT;                               // accumulator quads q = 2 t + ft of a hidden unit
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, gq = lane >> 4;
  const int n_my = (n_mt - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // >= 1 (grid <= n_mt)

  const float s_1 = scale_of(f.amax_in1), s_2 = scale_of(f.amax_in2);
  const float os1 = f.wsi1 / s_1, os2 = f.wsi2 / s_2;
  float amax1 = 0.f, amax2 = 0.f;
  const unsigned long long t_start = (ABL & 64) ? __builtin_amdgcn_s_memtime() : 0, r_start = (ABL & 64) ? __builtin_amdgcn_s_memrealtime() : 0;

  // ---- weight ring (as ffx.hip): wave w copies bytes [8 w KB, 8 w KB + 8 KB) of every slab as 8 LDS-DMA pieces of 1 KB ----
  const char* wsrc = reinterpret_cast<const char*>(f.Wstream) + wave * 8192 + lane * 16;
  int is_q = 0, is_g = 0;
  const char* cur_src = wsrc; unsigned cur_dst = 0;
  auto dma_begin = [&]() __attribute__((always_inline)) {
    cur_src = wsrc + (long)is_q * F6_SLAB;
    cur_dst = (unsigned)(uintptr_t)(smem + (is_g & (F6_R - 1)) * F6_SLAB + wave * 8192);
    is_q = is_q + 1 == F6_SLABS ? 0 : is_q + 1;
    ++is_g;
  };
#define F6_PIECE(C) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2" \
                                 :: "v"(cur_src + ((C) >> 2) * 4096), "s"(cur_dst + ((C) >> 2) * 4096), "n"(((C) & 3) * 1024) : "memory", "m0")
  auto dma_piece = [&](int cc) __attribute__((always_inline)) {      // cc: compile-time after unrolling
    switch (cc) { case 0: F6_PIECE(0); break; case 1: F6_PIECE(1); break; case 2: F6_PIECE(2); break; case 3: F6_PIECE(3); break;
                  case 4: F6_PIECE(4); break; case 5: F6_PIECE(5); break; case 6: F6_PIECE(6); break; default: F6_PIECE(7); break; }
  };
  auto issue_slab = [&]() __attribute__((always_inline)) {
    dma_begin();
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) dma_piece(cc);
  };

  // ---- consumer state ------------------------------------------------------------------------------------------------
  u32x4 F[3][4];                                            // fragment ring: macro-step m of a slab in F[(m + fo) % 3], two steps ahead of the MFMAs
  int g = 0;
  const char* rd = smem + lane * 16;
  auto read_macro = [&](u32x4 (&dst)[4], int slot, int m) __attribute__((always_inline)) {
    const char* p = rd + slot * F6_SLAB + m * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = *reinterpret_cast<const u32x4*>(p + i * 1024);
  };
  auto slab_top = [&](auto vm_c) __attribute__((always_inline)) {
    constexpr int VM = decltype(vm_c)::value;
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VM) : "memory");
    __builtin_amdgcn_s_barrier();
    dma_begin();
  };

  u32x4 XB[8][NT][2];                                       // the wave's tokens as B operand: [k32 step][token half][plane]
  f32x4 acc2[16][NT];                                       // 256 features x 32 tokens: [feature tile][token half]
  f32x4 acc1[2][BWD ? 1 : 2][NQ];                           // [unit parity][a, g (backward: d(hg))][quad q = 2 t + ft]
  u32x4 HB[BWD ? 2 * NT : NT][2];                           // the second product's B operand: [(da / dg,) token half][plane]
  f32x4 st1[BWD ? NQ : 1], st2[BWD ? NQ : 1];               // backward: the stash of the unit E works on next
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // The twelve MFMAs of a macro-step: fragments FB = [P hi, P lo, Q hi, Q lo]; per fragment pair the three product terms, small
  // ones first (lo x hi, hi x lo, hi x hi), the two token halves alternating (no MFMA reads the accumulator of the one before it).
  //   X0 / X1: the accumulators of pair P for token half 0 / 1, BP[t][plane] its B operand; Y0 / Y1, BQ likewise for pair Q.
#define F6_MM(ACC, FA, BB, Z) ACC = mm32(FA, BB, (Z) ? zero4 : ACC)

  // one slab = 8 macro-steps; the fragment reads run two macro-steps ahead of the MFMAs (ffx.hip)
  auto slab = [&](auto kind_c, auto fo_c, int par, auto side, auto vm_c) __attribute__((always_inline)) {
    constexpr int KIND = decltype(kind_c)::value, FO = decltype(fo_c)::value;
    slab_top(vm_c);
    const int slot = g & (F6_R - 1), nslot = (g + 1) & (F6_R - 1);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      u32x4 (&FB)[4] = F[(m + FO) % 3];
      u32x4 (&FN)[4] = F[(m + 2 + FO) % 3];
      const char* np = rd + (m < 6 ? slot * F6_SLAB + (m + 2) * 4096 : nslot * F6_SLAB + (m - 6) * 4096);
      // operands of this macro-step
      //   forward  P1A / P1B (ft = KIND): P = a tile, Q = g tile of k32 step m -> acc1[par][0 / 1][2 t + ft], B = XB[m]
      //   forward  P2: P / Q = feature tiles 2 m, 2 m + 1 of the unit's k32 step -> acc2[2 m / 2 m + 1][t], B = HB[t]
      //   backward P1: P / Q = feature tiles ft = 0 / 1 of k32 step m -> acc1[par][0][2 t + ft], B = XB[m]
      //   backward P2A / P2B: feature tile nt = 8 (KIND - P2A) + m, P = the da step, Q = the dg step -> acc2[nt][t], B = HB[2 w + t]
      constexpr bool P1 = KIND == Q_P1A || KIND == Q_P1B;
      const bool Z = P1 && m == 0;
      constexpr int T1 = NT - 1;                             // the second token half (NT = 1: none -- X1 / Y1 alias X0 / Y0 and are never used)
      f32x4* X0; f32x4* X1; f32x4* Y0; f32x4* Y1;
      const u32x4 (*BP)[2]; const u32x4 (*BQ)[2];
      if constexpr (!BWD) {
        if constexpr (P1) {
          X0 = &acc1[par][0][KIND]; X1 = &acc1[par][0][2 * T1 + KIND]; Y0 = &acc1[par][1][KIND]; Y1 = &acc1[par][1][2 * T1 + KIND];
          BP = XB[m]; BQ = XB[m];
        } else {
          X0 = &acc2[2 * m][0]; X1 = &acc2[2 * m][T1]; Y0 = &acc2[2 * m + 1][0]; Y1 = &acc2[2 * m + 1][T1];
          BP = HB; BQ = HB;
        }
      } else {
        if constexpr (P1) {
          X0 = &acc1[par][0][0]; X1 = &acc1[par][0][2 * T1]; Y0 = &acc1[par][0][1]; Y1 = &acc1[par][0][2 * T1 + 1];
          BP = XB[m]; BQ = XB[m];
        } else {
          const int nt = 8 * (KIND - Q_P2A) + m;
          X0 = &acc2[nt][0]; X1 = &acc2[nt][T1]; Y0 = X0; Y1 = X1;
          BP = HB; BQ = HB + NT;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      F6_MM(*X0, FB[1], BP[0][0], Z);
      __builtin_amdgcn_sched_barrier(0);
      dma_piece(m);
      __builtin_amdgcn_sched_barrier(0);
      // ---- one scheduling region: the four fragment reads, MFMAs 2..12 and this step's share of the elementwise work ----
#pragma unroll
      for (int i = 0; i < 4; ++i) FN[i] = *reinterpret_cast<const u32x4*>(np + i * 1024);
      if constexpr (NT == 2) {
      F6_MM(*X1, FB[1], BP[1][0], Z);
      F6_MM(*X0, FB[0], BP[0][1], false); F6_MM(*X1, FB[0], BP[1][1], false);
      F6_MM(*X0, FB[0], BP[0][0], false); F6_MM(*X1, FB[0], BP[1][0], false);
      F6_MM(*Y0, FB[3], BQ[0][0], Z); F6_MM(*Y1, FB[3], BQ[1][0], Z);
      F6_MM(*Y0, FB[2], BQ[0][1], false); F6_MM(*Y1, FB[2], BQ[1][1], false);
      F6_MM(*Y0, FB[2], BQ[0][0], false); F6_MM(*Y1, FB[2], BQ[1][0], false);
      } else {      // one token half: the P and Q chains alternate (no MFMA reads the accumulator of the one before it)
      F6_MM(*Y0, FB[3], BQ[0][0], Z);
      F6_MM(*X0, FB[0], BP[0][1], false); F6_MM(*Y0, FB[2], BQ[0][1], false);
      F6_MM(*X0, FB[0], BP[0][0], false); F6_MM(*Y0, FB[2], BQ[0][0], false);
      }
      side(m);
      // issue order hints: one fragment read per 16-cycle MFMA gap (a second one saturates the LDS array beside it), up to two
      // vector instructions per gap (an MFMA holds the vector issue for 8 of its 16 cycles), what is left behind the last MFMA
      if constexpr (NT == 1) {      // six MFMAs: a fragment read in four of the five gaps behind the first
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 12, 0);
      } else {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 12, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    ++g;
  };
  auto no_side = [](int) __attribute__((always_inline)) {};
  // the loop-carried accumulators stay where they are: see the header
  auto pin_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int t = 0; t < NT; ++t) asm volatile("" : "+a"(acc2[i][t]));
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int j = 0; j < (BWD ? 1 : 2); ++j)
#pragma unroll
        for (int q = 0; q < NQ; ++q) asm volatile("" : "+a"(acc1[p][j][q]));
  };

  // ---- prologue: b1 into LDS, first three slabs in flight --------------------------------------------------------------
  if (!BWD) {
    float* b1w = reinterpret_cast<float*>(smem + F6_B1);
    for (int i = tid; i < 512; i += 256) reinterpret_cast<f32x4*>(b1w)[i] = reinterpret_cast<const f32x4*>(f.b1)[i];
    reinterpret_cast<float*>(smem + F6_B2)[tid] = f.b2[tid];
  }
  {
    float* lns = reinterpret_cast<float*>(smem + F6_LN);
    if (tid < 64) reinterpret_cast<f32x4*>(lns)[tid] = reinterpret_cast<const f32x4*>(f.ln_g)[tid];
    else if (tid < 128 && f.ln_b) reinterpret_cast<f32x4*>(lns)[tid] = reinterpret_cast<const f32x4*>(f.ln_b)[tid - 64];
  }
  issue_slab(); issue_slab(); issue_slab();
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");          // slab 0 (my share)
  __syncthreads();                                           // (also publishes b1 in LDS)
  read_macro(F[0], 0, 0); read_macro(F[1], 0, 1);

  const float* b1s = reinterpret_cast<const float*>(smem + F6_B1);
  const float* lng = reinterpret_cast<const float*>(smem + F6_LN);

  for (int ti = 0; ti < n_my; ++ti) {
    const int mt = (int)blockIdx.x + ti * (int)gridDim.x;
    // NT = 2: tile mt = tokens 128 mt .., stash slot mt.  NT = 1: half tile mt of this launch = tokens tok0 + 64 mt .., the (mt & 1) half of slot slot0 + mt / 2
    float* stash_w = NT == 2 ? f.stash + (((long)mt * 32) * 4 + wave) * 2048 + lane * 4      // + unit * 8192 + (2 q + which) * 256
                             : f.stash + (((long)(f.slot0 + (mt >> 1)) * 32) * 4 + wave) * 2048 + lane * 4 + (mt & 1) * 1024;
    const long tok_w = NT == 2 ? (long)mt * 128 + wave * 32 + c : (long)f.tok0 + (long)mt * 64 + wave * 16 + c;      // + 16 t

    // ---- the wave's 32 tokens -> B-operand planes (forward: through LayerNorm-3) ---------------------------------------
    // lane (c, gq) holds k = 32 ks + 8 gq + i of the tokens 16 t + c
    {
      f32x4 xv[NT][16];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        long tok = tok_w + 16 * t;
        tok = tok < f.M ? tok : f.M - 1;
        const float* xrow = f.X + tok * 256 + 8 * gq;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          xv[t][2 * ks] = *reinterpret_cast<const f32x4*>(xrow + 32 * ks);
          xv[t][2 * ks + 1] = *reinterpret_cast<const f32x4*>(xrow + 32 * ks + 4);
        }
      }
      if (!BWD) {
        float sum[NT], ss[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          float a = 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i) a += (xv[t][i][0] + xv[t][i][1]) + (xv[t][i][2] + xv[t][i][3]);
          sum[t] = a;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) sum[t] = gsum(sum[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float mean = sum[t] * (1.f / 256.f);
          float a = 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = xv[t][i][e] - mean; a += d * d; }
          ss[t] = a;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) ss[t] = gsum(ss[t]);
        float mean_[NT], rstd_[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { mean_[t] = sum[t] * (1.f / 256.f); rstd_[t] = 1.f / sqrtf(ss[t] * (1.f / 256.f) + 1e-5f); }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int k = 32 * (i >> 1) + 8 * gq + 4 * (i & 1);
          const f32x4 gm = *reinterpret_cast<const f32x4*>(f.ln_g + k), bt = *reinterpret_cast<const f32x4*>(f.ln_b + k);      // (global: see ffx.hip)
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < NT; ++t) xv[t][i][e] = (xv[t][i][e] - mean_[t]) * rstd_[t] * gm[e] + bt[e];
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          amax_pin(amax1, xv[t][2 * ks][0], xv[t][2 * ks][1]); amax_pin(amax1, xv[t][2 * ks][2], xv[t][2 * ks][3]);
          amax_pin(amax1, xv[t][2 * ks + 1][0], xv[t][2 * ks + 1][1]); amax_pin(amax1, xv[t][2 * ks + 1][2], xv[t][2 * ks + 1][3]);
          split8(xv[t][2 * ks] * s_1, xv[t][2 * ks + 1] * s_1, XB[ks][t][0], XB[ks][t][1]);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc2[i][t] = zero4;

    auto stash_load = [&](int u) __attribute__((always_inline)) {                           // backward: prefetch unit u's stash
      if (BWD) {
        const float* p = stash_w + (long)u * 8192;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          st1[BWD ? q : 0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (2 * q) * 256));      // (read once)
          st2[BWD ? q : 0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (2 * q + 1) * 256));
        }
      }
    };
    // E(u): the elementwise step between the two products, on acc1[par], cut into work items that the slabs of a group take one
    // at a time (E_step) -- ffx.hip's, on the quads q = 2 t + ft
    f32x4 hq[BWD ? 2 * NQ : NQ];                             // forward: h quads; backward: [da quads | dg quads]
    f32x4 ba[2], bg[2];                                      // forward: b1 of quad q in [q & 1], read a stage ahead
    f32x4 qa[2], qg[2], qt[2], qp[2], s1q, s2q;              // the quad in progress [q & 1]: a, g, t = 1 / (1 + p |g|) then Phi(g), phi(g)
    auto bias_load = [&](int u, int q) __attribute__((always_inline)) {      // quad q = features 16 (q & 1) + 4 gq .. + 3 of unit u
      if constexpr (!BWD) {
        ba[q & 1] = *reinterpret_cast<const f32x4*>(b1s + (2 * u) * 32 + 16 * (q & 1) + 4 * gq);
        bg[q & 1] = *reinterpret_cast<const f32x4*>(b1s + (2 * u + 1) * 32 + 16 * (q & 1) + 4 * gq);
      }
    };
    bias_load(0, 0);
    auto geglu_stage = [&](int u, int par, int q, int st, int hf) __attribute__((always_inline)) {      // elements 2 hf, 2 hf + 1 of quad q
      if constexpr (!BWD) {
        if (st == 0) {
#pragma unroll
          for (int e = 2 * hf; e < 2 * hf + 2; ++e) {
            qa[q & 1][e] = fmaf(acc1[par][0][q][e], os1, ba[q & 1][e]);
            qg[q & 1][e] = fmaf(acc1[par][1][q][e], os1, bg[q & 1][e]);
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
          if (hf == 1) { if (q < NQ - 1) bias_load(u, q + 1); else bias_load(u < 31 ? u + 1 : 31, 0); }
        } else {
#pragma unroll
          for (int e = 2 * hf; e < 2 * hf + 2; ++e) {
            s1q[e] = qg[q & 1][e] * qt[q & 1][e];                            // gelu(g)
            s2q[e] = qa[q & 1][e] * fmaf(qg[q & 1][e], qp[q & 1][e], qt[q & 1][e]);   // a * gelu'(g)
            hq[q][e] = qa[q & 1][e] * s1q[e];                                // a * gelu(g)
          }
          amax_pin(amax2, hq[q][2 * hf], hq[q][2 * hf + 1]);
          if (!(ABL & 1)) { hq[q][2 * hf] *= s_2; hq[q][2 * hf + 1] *= s_2; }      // (ABL 1: the scale rides in the split, split2x)
          if (hf == 1) {
            float* p = stash_w + (long)u * 8192;
            __builtin_nontemporal_store(s1q, reinterpret_cast<f32x4*>(p + (2 * q) * 256));      // (non-temporal: ffx.hip)
            __builtin_nontemporal_store(s2q, reinterpret_cast<f32x4*>(p + (2 * q + 1) * 256));
          }
        }
      }
    };
    auto quad_b = [&](int par, int q) __attribute__((always_inline)) {
      if constexpr (BWD) {
        const f32x4 d = acc1[par][0][q] * os1;
        const f32x4 da = d * st1[q], dg = d * st2[q];
        amax_pin(amax2, da[0], da[1]); amax_pin(amax2, da[2], da[3]); amax_pin(amax2, dg[0], dg[1]); amax_pin(amax2, dg[2], dg[3]);
        if (!(ABL & 1)) { hq[q] = da * s_2; hq[NQ + q] = dg * s_2; } else { hq[q] = da; hq[NQ + q] = dg; }
      }
    };
    // B operand of token half t (backward: tt = 2 w + t): elements jj = 0..3 from the ft = 0 quad, 4..7 from the ft = 1 quad
    auto pack_half = [&](int t, int hf) __attribute__((always_inline)) {
      unsigned h0, h1, l0, l1;
      if (!(ABL & 1)) split4(hq[2 * t + hf], h0, h1, l0, l1);
      else { const f32x4 v = hq[2 * t + hf]; split2x(v[0], v[1], s_2, h0, l0); split2x(v[2], v[3], s_2, h1, l1); }
      HB[t][0][2 * hf] = h0; HB[t][0][2 * hf + 1] = h1; HB[t][1][2 * hf] = l0; HB[t][1][2 * hf + 1] = l1;
    };
    auto pack1 = [&](int tt) __attribute__((always_inline)) {
      if (!(ABL & 1)) split8(hq[2 * tt], hq[2 * tt + 1], HB[tt][0], HB[tt][1]);
      else { pack_half(tt, 0); pack_half(tt, 1); }
    };
    auto E_step = [&](int u, int par, int idx, int n, int pack_from) __attribute__((always_inline)) {
      if constexpr (!BWD) {
#pragma unroll
        for (int it = 0; it < 14 * NT; ++it) {               // q0: s0 h0, s0 h1, s1 h0, .. s2 h1; q1: ..; P0 h0, P0 h1; (NT = 2:) q2; q3; P1 h0, P1 h1
          const bool is_pack = (it >= 12 && it < 14) || it >= 26;
          int at = it * n / (14 * NT);
          if (is_pack && at < pack_from) at = pack_from;
          if (at > n - 1) at = n - 1;
          if (at != idx) continue;
          if (is_pack) pack_half(it >= 26 ? 1 : 0, it & 1);
          else { const int j = it < 12 ? it : it - 2; geglu_stage(u, par, j / 6, (j % 6) >> 1, j & 1); }
        }
      } else {
#pragma unroll
        for (int it = 0; it < 2 * NQ + 1; ++it) {            // the NQ quads, the next unit's stash, the NQ packs [da t.., dg t..]
          int at = it < NQ ? it * pack_from / NQ : (it == NQ ? pack_from - 1 : pack_from + (it - NQ - 1) * (n - pack_from) / NQ);
          if (at > n - 1) at = n - 1;
          if (at != idx) continue;
          if (it < NQ) quad_b(par, it);
          else if (it == NQ) { if (u + 1 < 32) stash_load(u + 1); }
          else pack1(it - NQ - 1);
        }
      }
    };

    // ---- the 32 hidden units, software-pipelined: P2(k - 1), P1(k + 1) and E(k) share a group (ffx.hip) ------------------
    using KA = std::integral_constant<int, Q_P1A>; using KB = std::integral_constant<int, Q_P1B>;
    using KC = std::integral_constant<int, Q_P2A>; using KD = std::integral_constant<int, Q_P2B>;
    using V8 = std::integral_constant<int, 8>; using V16 = std::integral_constant<int, 8 + 2 * NQ>;      // (the slab's 8 pieces + the 2 NQ stash loads of a unit)
    pin_acc();
    if constexpr (!BWD) {
      slab(KA{}, QO<0>{}, 0, no_side, V8{}); slab(KB{}, QO<2>{}, 0, no_side, V8{});
      slab(KA{}, QO<1>{}, 1, [&](int m) __attribute__((always_inline)) { E_step(0, 0, m, 16, 0); }, V8{});
      slab(KB{}, QO<0>{}, 1, [&](int m) __attribute__((always_inline)) { E_step(0, 0, 8 + m, 16, 0); }, V8{});
#pragma unroll 1
      for (int kk = 1; kk <= 29; kk += 2) {
        pin_acc();
#pragma unroll
        for (int o = 0; o < 2; ++o) {                        // k = kk + o: acc1 parity of E(k) is k & 1 = 1 - o
          const int k = kk + o, pe = 1 - o, pn = o;
          slab(KC{}, QO<2>{}, 0, [&](int m) __attribute__((always_inline)) { E_step(k, pe, m, 24, 8); }, V8{});
          slab(KA{}, QO<1>{}, pn, [&](int m) __attribute__((always_inline)) { E_step(k, pe, 8 + m, 24, 8); }, V8{});
          slab(KB{}, QO<0>{}, pn, [&](int m) __attribute__((always_inline)) { E_step(k, pe, 16 + m, 24, 8); }, V8{});
        }
      }
      pin_acc();
      slab(KC{}, QO<2>{}, 0, [&](int m) __attribute__((always_inline)) { E_step(31, 1, m, 9, 8); }, V8{});      // P2(30) with E(31)
      E_step(31, 1, 8, 9, 8);                                                                    // its packs, after P2(30)
      slab(KC{}, QO<1>{}, 0, no_side, V8{});                                                                    // P2(31)
    } else {
      stash_load(0);
      slab(KA{}, QO<0>{}, 0, no_side, V8{});
      slab(KA{}, QO<2>{}, 1, [&](int m) __attribute__((always_inline)) { E_step(0, 0, m, 8, 4); }, V8{});
#pragma unroll 1
      for (int kk = 1; kk <= 29; kk += 2) {
        pin_acc();
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const int k = kk + o, pe = 1 - o, pn = o;
          slab(KC{}, QO<1>{}, 0, [&](int m) __attribute__((always_inline)) { E_step(k, pe, m, 24, 16); }, V8{});
          slab(KD{}, QO<0>{}, 0, [&](int m) __attribute__((always_inline)) { E_step(k, pe, 8 + m, 24, 16); }, V8{});
          slab(KA{}, QO<2>{}, pn, [&](int m) __attribute__((always_inline)) { E_step(k, pe, 16 + m, 24, 16); }, V16{});   // (2 NQ stash loads + 8 pieces younger)
        }
      }
      pin_acc();
      slab(KC{}, QO<1>{}, 0, [&](int m) __attribute__((always_inline)) { E_step(31, 1, m, 20, 16); }, V8{});
      slab(KD{}, QO<0>{}, 0, [&](int m) __attribute__((always_inline)) { E_step(31, 1, 8 + m, 20, 16); }, V8{});
#pragma unroll
      for (int i = 16; i < 20; ++i) E_step(31, 1, i, 20, 16);
      slab(KC{}, QO<2>{}, 0, no_side, V8{}); slab(KD{}, QO<1>{}, 0, no_side, V8{});
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

    // ---- epilogue: lane (c, gq) holds features n = 16 nt + 4 gq + i of the tokens 16 t + c --------------------------------
    // (row addresses recomputed from an opaque copy of the tile index: ffx.hip)
    int mt_e = mt;
    asm volatile("" : "+s"(mt_e));
    long tok_e[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      tok_e[t] = (NT == 2 ? (long)mt_e * 128 + wave * 32 : (long)f.tok0 + (long)mt_e * 64 + wave * 16) + 16 * t + c;
      tok_e[t] = tok_e[t] < f.M ? tok_e[t] : f.M - 1;
    }
    if (!BWD) {
      const float* b2s = reinterpret_cast<const float*>(smem + F6_B2) + 4 * gq;
      // all 32 residual quads first, every store unconditional (tokens past M recompute and rewrite row M - 1 with the same bits)
      f32x4 rz[NT][16];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float* zrow = f.Z1 + tok_e[t] * 256 + 4 * gq;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) rz[t][nt] = *reinterpret_cast<const f32x4*>(zrow + 16 * nt);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float* orow = f.Y + tok_e[t] * 256 + 4 * gq;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
          const f32x4 b2 = *reinterpret_cast<const f32x4*>(b2s + 16 * nt);
          const f32x4 v = acc2[nt][t] * os2 + b2 + rz[t][nt];
          *reinterpret_cast<f32x4*>(orow + 16 * nt) = v;
        }
      }
    } else {
      // dz1 = dz + LNbwd(d(ln3); z1, gamma)   (rowops.hip, ln_bwd_kernel), one token half at a time
      const float* lgq = lng + 4 * gq;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float* zrow = f.Z1 + tok_e[t] * 256 + 4 * gq;
        const float* drow = f.X + tok_e[t] * 256 + 4 * gq;
        float* orow = f.Y + tok_e[t] * 256 + 4 * gq;
        f32x4 xz[16], add[16];
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) xz[nt] = *reinterpret_cast<const f32x4*>(zrow + 16 * nt);
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) add[nt] = *reinterpret_cast<const f32x4*>(drow + 16 * nt);
        float sum = 0.f;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) sum += (xz[nt][0] + xz[nt][1]) + (xz[nt][2] + xz[nt][3]);
        sum = gsum(sum);
        const float mean = sum * (1.f / 256.f);
        float ss = 0.f;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt)
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d = xz[nt][e] - mean; ss += d * d; }
        ss = gsum(ss);
        const float rstd = 1.f / sqrtf(ss * (1.f / 256.f) + 1e-5f);
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
          const f32x4 gm = *reinterpret_cast<const f32x4*>(lgq + 16 * nt);
          const f32x4 gv = acc2[nt][t] * os2 * gm;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xz[nt][e] = (xz[nt][e] - mean) * rstd;
            t1 += gv[e]; t2 += gv[e] * xz[nt][e];
          }
          acc2[nt][t] = gv;
        }
        t1 = gsum(t1); t2 = gsum(t2);
        const float m1 = t1 * (1.f / 256.f), m2 = t2 * (1.f / 256.f);
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (acc2[nt][t][e] - m1 - xz[nt][e] * m2) * rstd + add[nt][e];
          *reinterpret_cast<f32x4*>(orow + 16 * nt) = o;         // (unconditional: see the forward epilogue)
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");          // (ffx.hip: bound the operations in flight across the tile switch)
  }
#undef F6_PIECE
#undef F6_MM
  if ((ABL & 64) && f.stamps && lane == 0) {
    unsigned long long* o = f.stamps + ((long)blockIdx.x * 4 + wave) * 6;
    o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 0;
    o[4] = __builtin_amdgcn_s_memtime() - t_start;          // shader cycles of the whole kernel ..
    o[5] = __builtin_amdgcn_s_memrealtime() - r_start;      // .. over 100 MHz ticks: the clock it ran at
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no LDS-DMA may outlive the block

  // ---- maxima for the next evaluation's scales, range guard (as ffx.hip) -----------------------------------------------
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { amax1 = fmaxf(amax1, __shfl_xor(amax1, o)); amax2 = fmaxf(amax2, __shfl_xor(amax2, o)); }
  record_amax_block(f.amax_out1, amax1, reinterpret_cast<float*>(smem));
  record_amax_block(f.amax_out2, amax2, reinterpret_cast<float*>(smem) + 4);
  if (lane == 0) {
    if (f.range_flag) {
      if (!(amax1 * s_1 < 60000.f) || (amax1 > 0.f && amax1 * s_1 < 0.125f)) atomicMax(f.range_flag, f.site1 + 1);
      if (!(amax2 * s_2 < 60000.f) || (amax2 > 0.f && amax2 * s_2 < 0.125f)) atomicMax(f.range_flag, f.site2 + 1);
    }
  }
}

template <bool BWD, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void ffx16_kernel(FfxArgs f, int n_mt) { ffx16_body<BWD, ABL, 2>(f, n_mt); }
// half tiles (NT = 1): see ffx16_body
template <bool BWD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void ffx16h_kernel(FfxArgs f, int n_mt) { ffx16_body<BWD, 0, 1>(f, n_mt); }

// ---- weights -------------------------------------------------------------------------------------------------------------
// W [rows][cols] fp32 -> tmp [2 rows][cols / 2] such that launch_pack_h3(tmp) writes 16 x 32 fragments: its 32 x 16 fragment
// [R][s], lane 32 h + r, element j reads tmp[32 R + r][16 s + 8 h + j]; with r = 16 b + c that lane is lane 16 gq + c of the
// 16 x 16 x 32 operand for gq = 2 h + b, which must hold W'[16 R + c][32 s + 8 gq + j].  W' = W with its columns permuted:
//   perm 0: none (the first products: W1 in the [32 a | 32 g] tiling, W2^T)
//   perm 1: forward second product, W2 [256][1024]: per unit u, packed k = 8 gq + jj <- hidden 32 u + 16 (jj >> 2) + 4 gq + (jj & 3)
//           (the accumulator rows of the first product's quads 2 t, 2 t + 1)
//   perm 2: backward second product, W1^T [256][2048]: per unit u two k32 steps, w = 0 from the a-columns, w = 1 from the g-columns
__global__ void ffx16_gather_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols, int perm) {
  const long total = (long)rows * cols;
  const int oc = cols / 2;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long orow = idx / oc; const int ocol = (int)(idx - orow * oc);
    const long R = orow >> 5; const int b = (int)(orow >> 4) & 1, cc = (int)orow & 15;
    const int s = ocol >> 4, h = (ocol >> 3) & 1, j = ocol & 7;
    const long srow = 16 * R + cc;
    const int pcol = 32 * s + 16 * h + 8 * b + j;            // packed column: k32 step s, gq = 2 h + b, jj = j
    const int gq = (pcol >> 3) & 3, jj = pcol & 7;
    int scol = pcol;
    if (perm == 1) scol = (pcol & ~31) + 16 * (jj >> 2) + 4 * gq + (jj & 3);
    else if (perm == 2) { const int u = pcol >> 6, w = (pcol >> 5) & 1; scol = (w ? 1024 : 0) + 32 * u + 16 * (jj >> 2) + 4 * gq + (jj & 3); }
    out[idx] = in[srow * cols + scol];
  }
}
int ffx16_pack(const float* W, int rows, int cols, int perm, float scale, float* tmp, unsigned short* out, hipStream_t s) {
  RAMP_REQUIRE(W && tmp && out && rows % 16 == 0 && cols % 32 == 0 && perm >= 0 && perm <= 2 && (perm != 2 || cols == 2048), "ffx16_pack: bad shape");
  hipLaunchKernelGGL(ffx16_gather_kernel, dim3(1024), dim3(256), 0, s, W, tmp, rows, cols, perm);
  RAMP_HIP_CHECK(hipGetLastError());
  return launch_pack_h3(tmp, out, 2L * rows, cols / 2, scale, s);       // fragment [R][ks][plane]: ((R * (cols / 32) + ks) * 2 + plane) * 512 halves
}

// The weight stream of one direction: 96 slabs x 32 KB in ffx.hip's slab order; a slab = 8 macro-steps x 4 fragments [P hi, P lo, Q hi, Q lo].
//   forward : P1(u)x: macro m = k32 step m of feature tile ft = x: P = a rows (tile 4 u + ft), Q = g rows (tile 4 u + 2 + ft) of W1 (8 k32 steps)
//             P2(u) : macro m = feature tiles 2 m (P), 2 m + 1 (Q) of W2 (permuted), k32 step u (of 32)
//   backward: P1(u) : macro m = k32 step m of W2^T's row tiles 2 u (P), 2 u + 1 (Q) (8 k32 steps)
//             P2(u)x: macro m = feature tile 8 x + m of W1^T (permuted), k32 steps 2 u (P: da), 2 u + 1 (Q: dg) (of 64)
__global__ void ffx16_build_stream_kernel(const unsigned short* __restrict__ p1, const unsigned short* __restrict__ p2,
                                          unsigned short* __restrict__ out, int bwd) {
  const int q = blockIdx.x;                                  // slab
  int kind, u;
  if (!bwd) {
    if (q < 4) { kind = q & 1; u = q >> 1; }
    else if (q < 94) { const int k = 1 + (q - 4) / 3, j = (q - 4) - 3 * (k - 1); if (j == 0) { kind = Q_P2A; u = k - 1; } else { kind = j - 1; u = k + 1; } }
    else { kind = Q_P2A; u = q - 64; }
  } else {
    if (q < 2) { kind = Q_P1A; u = q; }
    else if (q < 92) { const int k = 1 + (q - 2) / 3, j = (q - 2) - 3 * (k - 1); if (j == 2) { kind = Q_P1A; u = k + 1; } else { kind = Q_P2A + j; u = k - 1; } }
    else { kind = Q_P2A + (q & 1); u = 30 + ((q - 92) >> 1); }
  }
  for (int i = threadIdx.x >> 6; i < 32; i += blockDim.x >> 6) {       // fragment i = 4 m + j of the slab
    const int m = i >> 2, j = i & 3, lane = threadIdx.x & 63;
    const int pq = j >> 1, plane = j & 1;
    long frag;                                               // fragment index [R][ks] in its packed array
    const unsigned short* base;
    if (!bwd) {
      if (kind <= Q_P1B) { base = p1; frag = (long)(4 * u + 2 * pq + kind) * 8 + m; }
      else { base = p2; frag = (long)(2 * m + pq) * 32 + u; }
    } else {
      if (kind == Q_P1A) { base = p1; frag = (long)(2 * u + pq) * 8 + m; }
      else { base = p2; frag = (long)(8 * (kind - Q_P2A) + m) * 64 + 2 * u + pq; }
    }
    reinterpret_cast<u32x4*>(out + ((long)q * 32 + i) * 512)[lane] = reinterpret_cast<const u32x4*>(base + (frag * 2 + plane) * 512)[lane];
  }
}
int ffx16_build_stream(const unsigned short* p1, const unsigned short* p2, unsigned short* out, bool bwd, hipStream_t s) {
  RAMP_REQUIRE(p1 && p2 && out, "ffx16_build_stream: null operand");
  hipLaunchKernelGGL(ffx16_build_stream_kernel, dim3(F6_SLABS), dim3(256), 0, s, p1, p2, out, bwd ? 1 : 0);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_ffx16(const FfxArgs& f, bool bwd, hipStream_t s) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  RAMP_REQUIRE(f.M > 0 && f.X && f.Y && f.Z1 && f.stash && f.Wstream && f.ln_g && (bwd || (f.ln_b && f.b1 && f.b2)), "ffx16: null operand");
  RAMP_REQUIRE(al16(f.X) && al16(f.Y) && al16(f.Z1) && al16(f.stash) && al16(f.Wstream) && al16(f.ln_g) && al16(f.ln_b) &&
               al16(f.b1) && al16(f.b2), "ffx16: operands must be 16-byte aligned");
  {   // rows past M are recomputed and rewritten from the inputs (unconditional stores): the output may alias none of them
    const size_t yb = (size_t)f.M * 256 * 4;
    RAMP_REQUIRE(!ranges_overlap(f.Y, yb, f.X, yb) && !ranges_overlap(f.Y, yb, f.Z1, yb), "ffx16: the output must not overlap X or z1 (no in-place use)");
  }
  const int n_mt = (f.M + 127) / 128, cus = device_cu_count();          // one 4-wave block per CU
  // tiling: full tiles of 128 tokens; the tiles of a last round that would leave half of the CUs idle -- and every tile of a launch that cannot fill half
  // of them -- run as two half tiles of 64 tokens each (ffx16h_kernel: ~0.7 tile times on twice the CUs).  Forward and backward launches of one M take
  // the same decision (the stash slots are addressed by it).
  RAMP_REQUIRE(f.half_mode >= 0 && f.half_mode <= 2 && (f.half_mode == 0 || f.ablate == 0), "ffx16: bad tiling mode");
  int n_full = n_mt;
  if (f.half_mode == 2) n_full = 0;
  else if (f.half_mode == 0 && f.ablate == 0) {
    const int r = n_mt % cus;
    if (2 * n_mt <= cus) n_full = 0;
    else if (n_mt > cus && r > 0 && 2 * r <= cus) n_full = n_mt - r;
  }
#define F6_GO(B, A) hipLaunchKernelGGL((ffx16_kernel<B, A>), dim3(std::min(n_full, cus)), dim3(256), F6_LDS, s, f, n_full)
  if (n_full > 0) {
    if (f.ablate == 0) { if (bwd) F6_GO(true, 0); else F6_GO(false, 0); }
    else if (f.ablate == 1) { if (bwd) F6_GO(true, 1); else F6_GO(false, 1); }
    else if (f.ablate == 64) { if (bwd) F6_GO(true, 64); else F6_GO(false, 64); }
    else RAMP_REQUIRE(false, "ffx16: ablation variant not built");
    RAMP_HIP_CHECK(hipGetLastError());
  }
#undef F6_GO
  if (n_full < n_mt) {
    FfxArgs h = f;
    h.tok0 = n_full * 128; h.slot0 = n_full;
    const int n_half = (f.M - h.tok0 + 63) / 64;
    if (bwd) hipLaunchKernelGGL((ffx16h_kernel<true>), dim3(std::min(n_half, cus)), dim3(256), F6_LDS, s, h, n_half);
    else hipLaunchKernelGGL((ffx16h_kernel<false>), dim3(std::min(n_half, cus)), dim3(256), F6_LDS, s, h, n_half);
    RAMP_HIP_CHECK(hipGetLastError());
  }
  return 0;
}

int init_ffx16_attributes() {
#define F6_ATTR(B, A) RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffx16_kernel<B, A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)F6_LDS))
  F6_ATTR(false, 0); F6_ATTR(true, 0); F6_ATTR(false, 1); F6_ATTR(true, 1); F6_ATTR(false, 64); F6_ATTR(true, 64);
#undef F6_ATTR
  RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffx16h_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)F6_LDS));
  RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffx16h_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)F6_LDS));
  return 0;
}

}  // namespace ramp

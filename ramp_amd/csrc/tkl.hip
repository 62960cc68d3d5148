// Token-owning linear layers with K = 256: LayerNorm-1 -> QKV, the attention output projection (+ bias, + the block's
// cross-attention constant per row variant, + residual) and its input gradient d(o)  (layers_attention_mini.py:130-149:
// x = attn1(norm1(x)) + x, and CrossAttention.to_q/to_k/to_v/to_out, :60-120).
//
// Same dataflow as ffx.hip -- a wave owns 32 tokens, their (LayerNorm'd) rows live in registers as two scaled fp16 planes for
// the whole tile, products are computed transposed (D^T[feature][token] = W X^T, weight fragment = MFMA A operand), the weights
// stream through an LDS ring filled by LDS-DMA -- with three differences that the single product allows:
//   * the result of 32 output features is complete after 48 MFMAs and leaves at once (scale, bias, residual, 16-byte stores
//     from the accumulator layout: a token's 32 features = one 128-byte line after the four quads), interleaved with the
//     NEXT 32 features' MFMAs -- the write burst that costs the tile kernels a third of their time on K = 256 shapes
//      is spread over the whole tile;
//   * 128 operand registers + 2 x 16 accumulators fit 256 VGPRs: TWO blocks per CU (two waves per SIMD), each with its own
//     4 x 16 KB ring, so one block's prologue / stores / DMA issue / barrier overlap the other's MFMAs;
//   * the packed fp16 planes of the tile kernels ([row / 32][k / 16][plane][lane][8], launch_pack_h3) ARE the weight stream:
//     32 KB per 32 features, consumed in two 16 KB slabs (k16 steps 0-7, 8-15).
// LayerNorm runs on the registers the tile holds anyway (ln_fwd and its round trip disappear from the fp16x3 evaluations).
#include "common.h"
#include "tokmma.h"

#include <algorithm>
#include <type_traits>

namespace ramp {

namespace {
constexpr int TK_SLAB = 16 * 1024;                      // 4 macro-steps x 4 fragments x 1 KB
constexpr int TK_R = 4;
constexpr int TK_BIAS = TK_R * TK_SLAB;                 // bias (<= 768 floats)
constexpr int TK_LN = TK_BIAS + 768 * 4;                // gamma | beta
constexpr int TK_RB = TK_LN + 512 * 4;                  // row-variant bias rows (<= 4 x 256)
constexpr size_t TK_LDS = (size_t)TK_RB + 4 * 256 * 4;
static_assert(2 * TK_LDS <= 160 * 1024, "two blocks per CU");
}  // namespace

// EPI: bit 0 residual, bit 1 row-variant bias.  ABL (diagnostic, ramp_bench_gemm): 1 no LDS-DMA, 2 no stores, 4 one block per CU, 8 no barrier
template <bool LN, int EPI, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void tkl_kernel(TklArgs a, int n_mt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int n_my = (n_mt - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // >= 1 (grid <= n_mt)
  const int nblk = a.N >> 5, spt = 2 * nblk;                // 32-feature blocks, slabs per tile

  const float s_in = scale_of(a.amax_in);
  const float os = a.wsi / s_in;
  float amax = 0.f;

  // ---- weight ring (see ffx.hip for the inline-asm LDS-DMA): wave w copies bytes [4 w KB, 4 w KB + 4 KB) of every slab ------
  const char* wsrc = reinterpret_cast<const char*>(a.W) + wave * 4096 + lane * 16;
  int is_q = 0, is_g = 0;
  const char* cur_src = wsrc; unsigned cur_dst = 0;
  auto dma_begin = [&]() __attribute__((always_inline)) {
    cur_src = wsrc + (long)is_q * TK_SLAB;
    cur_dst = (unsigned)(uintptr_t)(smem + (is_g & (TK_R - 1)) * TK_SLAB + wave * 4096);
    is_q = is_q + 1 == spt ? 0 : is_q + 1;
    ++is_g;
  };
#define TK_PIECE(C) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2" \
                                 :: "v"(cur_src), "s"(cur_dst), "n"((C) * 1024) : "memory", "m0")
  auto piece_raw = [&](int c) __attribute__((always_inline)) {
    switch (c) { case 0: TK_PIECE(0); break; case 1: TK_PIECE(1); break; case 2: TK_PIECE(2); break; default: TK_PIECE(3); break; }
  };
  auto dma_piece = [&](int c) __attribute__((always_inline)) { if (!(ABL & 1)) piece_raw(c); };
  auto issue_slab = [&]() __attribute__((always_inline)) {
    dma_begin();
#pragma unroll
    for (int c = 0; c < 4; ++c) piece_raw(c);
  };

  u32x4 F[2][4];                                            // fragments of macro-step m in F[m & 1], read one step ahead
  int g = 0;
  const char* rd = smem + lane * 16;
  u32x4 XB[16][2];                                          // the wave's tokens as B operand: [k16 step][plane]
  f32x16 acc[2];
  const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  // top of slab g: my pieces of slab g + 1 have landed (VM = the vector-memory operations this wave issued after them, an
  // exact count: waiting for fewer would also wait for the previous slab's stores), barrier -> slab g + 1 certified for
  // everybody and the slot of slab g - 1 free -> slab g + 3 goes there, one piece per macro-step
  auto slab = [&](auto half_c, auto vm_c, f32x16& ac, auto side) __attribute__((always_inline)) {
    constexpr int HALF = decltype(half_c)::value, VM = decltype(vm_c)::value;
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VM) : "memory");
    if (!(ABL & 8)) __builtin_amdgcn_s_barrier();
    dma_begin();
    const int slot = g & (TK_R - 1), nslot = (g + 1) & (TK_R - 1);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      u32x4 (&FB)[4] = F[m & 1];
      u32x4 (&FN)[4] = F[(m + 1) & 1];
      const char* np = rd + (m < 3 ? slot * TK_SLAB + (m + 1) * 4096 : nslot * TK_SLAB);
      const int s = 2 * (4 * HALF + m);
      __builtin_amdgcn_sched_barrier(0);
      ac = mfma16(FB[1], XB[s][0], (HALF == 0 && m == 0) ? zero16 : ac);
      __builtin_amdgcn_sched_barrier(0);
      dma_piece(m);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) FN[i] = *reinterpret_cast<const u32x4*>(np + i * 1024);
      ac = mfma16(FB[0], XB[s][1], ac); ac = mfma16(FB[0], XB[s][0], ac);
      ac = mfma16(FB[3], XB[s + 1][0], ac);
      ac = mfma16(FB[2], XB[s + 1][1], ac); ac = mfma16(FB[2], XB[s + 1][0], ac);
      side(m);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x402, 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 12, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    ++g;
  };
  auto no_side = [](int) __attribute__((always_inline)) {};

  // ---- prologue: tables into LDS, three slabs in flight ----------------------------------------------------------------------
  {
    float* bs = reinterpret_cast<float*>(smem + TK_BIAS);
    for (int i = tid; i < a.N; i += 256) bs[i] = a.bias ? a.bias[i] : 0.f;
    float* lns = reinterpret_cast<float*>(smem + TK_LN);
    if (LN) { lns[tid] = a.ln_g[tid]; lns[256 + tid] = a.ln_b[tid]; }
    if (EPI & 2) {
      float* rbs = reinterpret_cast<float*>(smem + TK_RB);
      for (int v = 0; v < a.n_var; ++v) rbs[v * 256 + tid] = a.rowbias[(long)v * a.rb_stride + tid];
    }
  }
  issue_slab(); issue_slab(); issue_slab();
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // slab 0 (my share)
  __syncthreads();                                           // (also publishes the tables)
#pragma unroll
  for (int i = 0; i < 4; ++i) F[0][i] = *reinterpret_cast<const u32x4*>(rd + i * 1024);

  const float* bs = reinterpret_cast<const float*>(smem + TK_BIAS) + 4 * h;
  const float* lng = reinterpret_cast<const float*>(smem + TK_LN);
  const float* lnb = lng + 256;
  using H0 = std::integral_constant<int, 0>; using H1 = std::integral_constant<int, 1>;
  using V4 = std::integral_constant<int, 4>; using V8 = std::integral_constant<int, 8>;       // 4 pieces + 4 stores of the slab before
  using VB = std::integral_constant<int, 4 + ((EPI & 1) ? 4 : 0)>;

  for (int ti = 0; ti < n_my; ++ti) {
    const int mt = (int)blockIdx.x + ti * (int)gridDim.x;
    const long tok = (long)mt * 128 + wave * 32 + r;
    // tokens past M work on (and rewrite, with the same bits) row M - 1: every lane issues every load and store, which the
    // exact vmcnt constants of the slab tops rely on
    const long tokc = tok < a.M ? tok : a.M - 1;

    // ---- the wave's 32 tokens -> B-operand planes (through LayerNorm) --------------------------------------------------------
    {
      const float* xrow = a.X + tokc * 256 + 8 * h;          // lane (r, h) holds k = 16 s + 8 h + i of token r
      f32x4 xv[32];
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // vmcnt is a 6-bit counter: the previous tile's last stores + pieces + these 32 loads stay below 63
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        xv[2 * s] = *reinterpret_cast<const f32x4*>(xrow + 16 * s);
        xv[2 * s + 1] = *reinterpret_cast<const f32x4*>(xrow + 16 * s + 4);
      }
      if (LN) {
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
          const f32x4 gm = *reinterpret_cast<const f32x4*>(lng + k), bt = *reinterpret_cast<const f32x4*>(lnb + k);
#pragma unroll
          for (int e = 0; e < 4; ++e) xv[i][e] = (xv[i][e] - mean) * rstd * gm[e] + bt[e];
        }
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        amax = amax4(xv[2 * s], amax); amax = amax4(xv[2 * s + 1], amax);
        split8(xv[2 * s] * s_in, xv[2 * s + 1] * s_in, XB[s][0], XB[s][1]);
      }
    }
    float* yrow = a.Y + tokc * a.ldy + 4 * h;
    const float* rrow = (EPI & 1) ? a.resid + tokc * a.ldr + 4 * h : nullptr;
    const float* rbp = nullptr;
    if (EPI & 2) rbp = reinterpret_cast<const float*>(smem + TK_RB) + a.rowvar[a.row0 + (int)(tokc / a.L)] * 256 + 4 * h;

    // epilogue of feature block pb (accumulator P) in two parts: the residual quads are requested during the first slab
    // of the next block, the four quads leave one per macro-step of its second slab
    f32x4 rz[4];
    auto epi_load = [&](int pb) __attribute__((always_inline)) {
      if (EPI & 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) rz[q] = *reinterpret_cast<const f32x4*>(rrow + 32 * pb + 8 * q);
      }
    };
    auto epi_quad = [&](const f32x16& P, int pb, int q) __attribute__((always_inline)) {
      f32x4 v = quad(P, q) * os + *reinterpret_cast<const f32x4*>(bs + 32 * pb + 8 * q);
      if (EPI & 2) v += *reinterpret_cast<const f32x4*>(rbp + 32 * pb + 8 * q);
      if (EPI & 1) v += rz[q];
      if (!(ABL & 2) || tok < 0) *reinterpret_cast<f32x4*>(yrow + 32 * pb + 8 * q) = v;      // (unconditional: see tokc)
    };

    slab(H0{}, V4{}, acc[0], no_side); slab(H1{}, V4{}, acc[0], no_side);
    if (nblk > 1) {
      slab(H0{}, V4{}, acc[1], [&](int m) __attribute__((always_inline)) { if (m == 0) epi_load(0); });
      slab(H1{}, VB{}, acc[1], [&](int m) __attribute__((always_inline)) { epi_quad(acc[0], 0, m); });
    }
#pragma unroll 1
    for (int nb = 2; nb < nblk; nb += 2) {
      slab(H0{}, V8{}, acc[0], [&](int m) __attribute__((always_inline)) { if (m == 0) epi_load(nb - 1); });
      slab(H1{}, VB{}, acc[0], [&](int m) __attribute__((always_inline)) { epi_quad(acc[1], nb - 1, m); });
      if (nb + 1 < nblk) {
        slab(H0{}, V8{}, acc[1], [&](int m) __attribute__((always_inline)) { if (m == 0) epi_load(nb); });
        slab(H1{}, VB{}, acc[1], [&](int m) __attribute__((always_inline)) { epi_quad(acc[0], nb, m); });
      }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    epi_load(nblk - 1);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    if ((nblk - 1) & 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) epi_quad(acc[1], nblk - 1, q);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) epi_quad(acc[0], nblk - 1, q);
    }
  }
#undef TK_PIECE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no LDS-DMA may outlive the block

#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
  if (lane == 0) {
    if (a.amax_out) atomicMax(reinterpret_cast<unsigned*>(a.amax_out), __builtin_bit_cast(unsigned, amax));
    if (a.range_flag && (!(amax * s_in < 60000.f) || (amax > 0.f && amax * s_in < 0.125f))) atomicMax(a.range_flag, a.site + 1);
  }
}

int launch_tkl(const TklArgs& a, hipStream_t s) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  RAMP_REQUIRE(a.M > 0 && a.N >= 32 && a.N <= 768 && a.N % 32 == 0 && a.X && a.Y && a.W, "tkl: bad operand");
  RAMP_REQUIRE(al16(a.X) && al16(a.Y) && al16(a.W) && al16(a.resid) && a.ldy % 4 == 0 && a.ldr % 4 == 0, "tkl: operands must be 16-byte aligned");
  RAMP_REQUIRE(!a.rowbias || (a.rowvar && a.N == 256 && a.n_var >= 1 && a.n_var <= 4 && a.L >= 1 && a.resid), "tkl: row-variant bias needs N = 256, <= 4 variants, a residual");
  RAMP_REQUIRE(!a.ln_g == !a.ln_b, "tkl: LayerNorm needs gamma and beta");
  const int n_mt = (a.M + 127) / 128;
  const int nb = std::min(n_mt, 512);                        // two 4-wave blocks per CU
  const bool ln = a.ln_g != nullptr;
  const int epi = (a.resid ? 1 : 0) | (a.rowbias ? 2 : 0);
#define TK_GO(LNV, E, A) hipLaunchKernelGGL((tkl_kernel<LNV, E, A>), dim3((A & 4) ? std::min(n_mt, 256) : nb), dim3(256), TK_LDS + ((A & 4) ? 40 * 1024 : 0), s, a, n_mt)
#define TK_ABL(A) else if (a.ablate == A && ln && epi == 0) TK_GO(true, 0, A); else if (a.ablate == A && !ln && epi == 1) TK_GO(false, 1, A); \
                  else if (a.ablate == A && !ln && epi == 0) TK_GO(false, 0, A);
  if (a.ablate == 0) {
    if (ln && epi == 0) TK_GO(true, 0, 0);
    else if (!ln && epi == 0) TK_GO(false, 0, 0);
    else if (!ln && epi == 1) TK_GO(false, 1, 0);
    else if (!ln && epi == 3) TK_GO(false, 3, 0);
    else RAMP_REQUIRE(false, "tkl: variant not built");
  }
  TK_ABL(1) TK_ABL(2) TK_ABL(4) TK_ABL(8) TK_ABL(3)
  else RAMP_REQUIRE(false, "tkl: ablation variant not built");
#undef TK_ABL
#undef TK_GO
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

int init_tkl_attributes() {
#define TK_ATTR(LNV, E, A) RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tkl_kernel<LNV, E, A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TK_LDS + ((A & 4) ? 40 * 1024 : 0)))
  TK_ATTR(true, 0, 0); TK_ATTR(false, 0, 0); TK_ATTR(false, 1, 0); TK_ATTR(false, 3, 0);
#define TK_ATTR3(A) TK_ATTR(true, 0, A); TK_ATTR(false, 1, A); TK_ATTR(false, 0, A)
  TK_ATTR3(1); TK_ATTR3(2); TK_ATTR3(4); TK_ATTR3(8); TK_ATTR3(3);
#undef TK_ATTR3
#undef TK_ATTR
  return 0;
}

}  // namespace ramp

// Token-owning linear layers with K = 256: LayerNorm-1 -> QKV, the attention output projection (+ bias, + the block's
// cross-attention constant per row variant, + residual) and its input gradient d(o)  (layers_attention_mini.py:130-149:
// x = attn1(norm1(x)) + x, and CrossAttention.to_q/to_k/to_v/to_out, :60-120).
//
// Same dataflow as ffx.hip -- one 4-wave block per CU, a wave owns 32 tokens, their (LayerNorm'd) rows live in registers as
// two scaled fp16 planes for the whole tile, products are computed transposed (D^T[feature][token] = W X^T, weight fragment =
// MFMA A operand), the weights stream through a 4 x 32 KB LDS ring filled by LDS-DMA, one barrier per slab -- with what the
// single product allows:
//   * a slab IS a block of 32 output features (32 x 256 weights = 32 KB of fragments), complete after its 48 MFMAs: it
//     leaves during the NEXT slab (wave-private LDS transpose -> scale, bias, row-variant constant, residual -> 16-byte
//     stores that cover eight tokens x one full 128-byte line per instruction), so the write burst that costs the tile
//     kernels a third of their time on K = 256 shapes is spread over the whole tile.  (Storing from the
//     accumulator layout directly -- 32-byte pieces, or with the operands swapped one dword per lane -- measured 40-60 %
//     slower: the store path charges per instruction and per partial line.);
//   * the next tile's rows are fetched into registers during the current tile's first eight slabs (no exposed load latency
//     at the tile switch);
//   * the packed fp16 planes of the tile kernels ([row / 32][k / 16][plane][lane][8], launch_pack_h3) ARE the weight stream.
// LayerNorm runs on the registers the tile holds anyway (ln_fwd and its round trip disappear from the fp16x3 evaluations).
#include "args_token.h"
#include "tokmma.h"

#include <algorithm>
#include <type_traits>

namespace ramp {

namespace {
constexpr int TK_SLAB = 32 * 1024;                      // 8 macro-steps x 4 fragments x 1 KB = 32 output features
constexpr int TK_R = 4;
constexpr int TK_TROW = 36;                             // floats per token row of the transpose scratch (32 + pad)
constexpr int TK_T = TK_R * TK_SLAB;                    // per wave 32 x 36 floats
constexpr int TK_BIAS = TK_T + 4 * 32 * TK_TROW * 4;    // bias (<= 768 floats)
constexpr int TK_LN = TK_BIAS + 768 * 4;                // gamma | beta
constexpr int TK_RB = TK_LN + 512 * 4;                  // row-variant bias rows (<= 4 x 256)
constexpr size_t TK_LDS = (size_t)TK_RB + 4 * 256 * 4;
static_assert(TK_LDS <= 160 * 1024, "LDS budget");
}  // namespace

// EPI: bit 0 residual, bit 1 row-variant bias.  ABL (diagnostic, ramp_bench_gemm): 1 no LDS-DMA, 2 no stores, 8 no barrier
template <bool LN, int EPI, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void tkl_kernel(TklArgs a, int n_mt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int n_my = (n_mt - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // >= 1 (grid <= n_mt)
  const int nblk = a.N >> 5;                                // 32-feature blocks = slabs per tile

  const float s_in = scale_of(a.amax_in);
  const float os = a.wsi / s_in;
  float amax = 0.f;

  // ---- weight ring (see ffx.hip): wave w copies bytes [8 w KB, 8 w KB + 8 KB) of every slab as 8 LDS-DMA pieces -------------
  const char* wsrc = reinterpret_cast<const char*>(a.W) + wave * 8192 + lane * 16;
  int is_q = 0, is_g = 0;
  const char* cur_src = wsrc; unsigned cur_dst = 0;
  auto dma_begin = [&]() __attribute__((always_inline)) {
    cur_src = wsrc + (long)is_q * TK_SLAB;
    cur_dst = (unsigned)(uintptr_t)(smem + (is_g & (TK_R - 1)) * TK_SLAB + wave * 8192);
    is_q = is_q + 1 == nblk ? 0 : is_q + 1;
    ++is_g;
  };
#define TK_PIECE(C) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2" \
                                 :: "v"(cur_src + ((C) >> 2) * 4096), "s"(cur_dst + ((C) >> 2) * 4096), "n"(((C) & 3) * 1024) : "memory", "m0")
  auto piece_raw = [&](int c) __attribute__((always_inline)) {
    switch (c) { case 0: TK_PIECE(0); break; case 1: TK_PIECE(1); break; case 2: TK_PIECE(2); break; case 3: TK_PIECE(3); break;
                 case 4: TK_PIECE(4); break; case 5: TK_PIECE(5); break; case 6: TK_PIECE(6); break; default: TK_PIECE(7); break; }
  };
  auto dma_piece = [&](int c) __attribute__((always_inline)) { if (!(ABL & 1)) piece_raw(c); };
  auto issue_slab = [&]() __attribute__((always_inline)) {
    dma_begin();
#pragma unroll
    for (int c = 0; c < 8; ++c) piece_raw(c);
  };

  u32x4 F[4][4];                                            // fragments of macro-step m in F[m & 3], read two steps ahead
  int g = 0;
  const char* rd = smem + lane * 16;
  u32x4 XB[16][2];                                          // the wave's tokens as B operand: [k16 step][plane]
  f32x4 xn[32];                                             // the NEXT tile's rows, raw
  f32x16 acc[2];
  const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  // one slab = 8 macro-steps = 48 MFMAs into one accumulator.  Top of slab g: my pieces of slab g + 1 have landed (the only
  // younger LDS-DMA are my 8 pieces of slab g + 2; the side work's loads and stores of slab g - 1 are younger too, so the wait
  // covers some of them: measured a few tens of cycles in ffx.hip), barrier -> slab g + 1 certified, slot of slab g - 1 free
  auto slab = [&](f32x16& ac, auto side) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (!(ABL & 8)) __builtin_amdgcn_s_barrier();
    dma_begin();
    const int slot = g & (TK_R - 1), nslot = (g + 1) & (TK_R - 1);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      u32x4 (&FB)[4] = F[m & 3];
      u32x4 (&FN)[4] = F[(m + 2) & 3];
      const char* np = rd + (m < 6 ? slot * TK_SLAB + (m + 2) * 4096 : nslot * TK_SLAB + (m - 6) * 4096);
      const int s = 2 * m;
      __builtin_amdgcn_sched_barrier(0);
      ac = mfma16(FB[1], XB[s][0], m == 0 ? zero16 : ac);
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

  // rows of tile mt -> xn (lane (r, h) holds k = 16 s + 8 h + i of token r); j: a quarter of the 32 loads
  auto x_load = [&](int mt, int j) __attribute__((always_inline)) {
    long tok = (long)mt * 128 + wave * 32 + r;
    tok = tok < a.M ? tok : a.M - 1;                        // (rows past M recompute row M - 1)
    const float* xrow = a.X + tok * 256 + 8 * h;
#pragma unroll
    for (int s = 2 * j; s < 2 * j + 2; ++s) {
      xn[2 * s] = *reinterpret_cast<const f32x4*>(xrow + 16 * s);
      xn[2 * s + 1] = *reinterpret_cast<const f32x4*>(xrow + 16 * s + 4);
    }
  };

  // ---- prologue: tables into LDS, the first tile's rows and three slabs in flight --------------------------------------------
  {
    float* bsw = reinterpret_cast<float*>(smem + TK_BIAS);
    for (int i = tid; i < a.N; i += 256) bsw[i] = a.bias ? a.bias[i] : 0.f;
    float* lns = reinterpret_cast<float*>(smem + TK_LN);
    if (LN) { lns[tid] = a.ln_g[tid]; lns[256 + tid] = a.ln_b[tid]; }
    if (EPI & 2) {
      float* rbs = reinterpret_cast<float*>(smem + TK_RB);
      for (int v = 0; v < a.n_var; ++v) rbs[v * 256 + tid] = a.rowbias[(long)v * a.rb_stride + tid];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) x_load((int)blockIdx.x, j);
  issue_slab(); issue_slab(); issue_slab();
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");          // slab 0 (my share; the rows are older)
  __syncthreads();                                           // (also publishes the tables)
#pragma unroll
  for (int i = 0; i < 4; ++i) { F[0][i] = *reinterpret_cast<const u32x4*>(rd + i * 1024); F[1][i] = *reinterpret_cast<const u32x4*>(rd + 4096 + i * 1024); }

  const int l8 = lane >> 3, c4 = 4 * (lane & 7);            // store layout: lane -> token 8 j + l8 of the wave, features c4 .. c4 + 3 of the block
  float* tw = reinterpret_cast<float*>(smem + TK_T) + wave * 32 * TK_TROW;
  float* tw_w = tw + r * TK_TROW + 4 * h;                   // accumulator layout: token r, features 8 q + 4 h ..
  const float* tw_r = tw + l8 * TK_TROW + c4;               // + 8 j rows
  const float* bs = reinterpret_cast<const float*>(smem + TK_BIAS) + c4;
  const float* lng = reinterpret_cast<const float*>(smem + TK_LN);
  const float* lnb = lng + 256;

  for (int ti = 0; ti < n_my; ++ti) {
    const int mt = (int)blockIdx.x + ti * (int)gridDim.x;
    const int mt_next = ti + 1 < n_my ? mt + (int)gridDim.x : mt;      // (last tile: re-reads its own rows, unused)

    // ---- the wave's 32 tokens -> B-operand planes (through LayerNorm) --------------------------------------------------------
    {
      if (LN) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) sum += (xn[i][0] + xn[i][1]) + (xn[i][2] + xn[i][3]);
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.f / 256.f);
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d = xn[i][e] - mean; ss += d * d; }
        ss += __shfl_xor(ss, 32);
        const float rstd = 1.f / sqrtf(ss * (1.f / 256.f) + 1e-5f);
        // one LDS base per tile (opaque) + immediate offsets: as `lng + k` hipcc kept 32 separate address registers alive across
        // the tile loop, spilled them and reloaded each behind an s_waitcnt vmcnt(0)
        unsigned lgh = (unsigned)(uintptr_t)(lng + 8 * h);
        asm volatile("" : "+v"(lgh));
        typedef __attribute__((address_space(3))) const char* lds_cptr_t;      // (an LDS pointer: 32 bits; a generic pointer rebuilt from them would be a flat load)
        typedef __attribute__((address_space(3))) const f32x4* lds_f4ptr_t;
        const lds_cptr_t lgp = (lds_cptr_t)lgh;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          const int k = 16 * (i >> 1) + 4 * (i & 1);
          const f32x4 gm = *(lds_f4ptr_t)(lgp + 4 * k), bt = *(lds_f4ptr_t)(lgp + 4 * (256 + k));
#pragma unroll
          for (int e = 0; e < 4; ++e) xn[i][e] = (xn[i][e] - mean) * rstd * gm[e] + bt[e];
          if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (keeps hipcc from hoisting all 64 table reads: 256 registers)
        }
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        amax_pin(amax, xn[2 * s][0], xn[2 * s][1]); amax_pin(amax, xn[2 * s][2], xn[2 * s][3]);      // (pinned: see tokmma.h)
        amax_pin(amax, xn[2 * s + 1][0], xn[2 * s + 1][1]); amax_pin(amax, xn[2 * s + 1][2], xn[2 * s + 1][3]);
        split8(xn[2 * s] * s_in, xn[2 * s + 1] * s_in, XB[s][0], XB[s][1]);
        if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
    // rows this lane stores (and reads the residual of): token 8 j + l8 of the wave, clamped (rows past M rewrite row M - 1
    // with the same bits: every lane issues every load and store)
    // (addresses are formed where they are used, from a tile index hipcc cannot see through: as per-tile arrays of row
    // pointers they were spilled and every slab reloaded them from scratch behind an s_waitcnt vmcnt(0))
    int rbo[4] = {0, 0, 0, 0};                              // LDS offset of each row's variant constants
    if (EPI & 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        long t = (long)mt * 128 + wave * 32 + 8 * j + l8;
        t = t < a.M ? t : a.M - 1;
        rbo[j] = a.rowvar[a.row0 + (int)(t / a.L)] * 256 + c4;
      }
    }
    auto row_of = [&](int j) __attribute__((always_inline)) {
      int mt_o = mt; asm volatile("" : "+s"(mt_o));
      long t = (long)mt_o * 128 + wave * 32 + 8 * j + l8;
      return t < a.M ? t : a.M - 1;
    };

    // epilogue of feature block pb (accumulator P), one step per macro-step of the next slab:
    // 0 residual requested | 1 accumulators -> transpose scratch | 2 read back token-major | 3 bias, constants | 4..7 one store each
    f32x4 rz[4], tv[4], bq, rbq[4];
    auto epi = [&](const f32x16& P, int pb, int st) __attribute__((always_inline)) {
      if (st == 0) {
        if (EPI & 1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) rz[j] = *reinterpret_cast<const f32x4*>(a.resid + row_of(j) * a.ldr + c4 + 32 * pb);
        }
      } else if (st == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(tw_w + 8 * q) = quad(P, q);
      } else if (st == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) tv[j] = *reinterpret_cast<const f32x4*>(tw_r + 8 * j * TK_TROW);
      } else if (st == 3) {
        bq = *reinterpret_cast<const f32x4*>(bs + 32 * pb);
        if (EPI & 2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) rbq[j] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(smem + TK_RB) + rbo[j] + 32 * pb);
        }
      } else {
        const int j = st - 4;
        f32x4 v = tv[j] * os + bq;
        if (EPI & 2) v += rbq[j];
        if (EPI & 1) v += rz[j];
        if (!(ABL & 2) || mt < 0) *reinterpret_cast<f32x4*>(a.Y + row_of(j) * a.ldy + c4 + 32 * pb) = v;
      }
    };
    auto no_side = [](int) __attribute__((always_inline)) {};

    // the tile's first eight slabs unrolled (each also fetches an eighth of the next tile's rows), the rest in pairs
    slab(acc[0], [&](int m) __attribute__((always_inline)) { if (m == 0) x_load(mt_next, 0); });
#define TK_EARLY(NB)                                                                                                                     \
    if (NB < nblk) slab(acc[NB & 1], [&](int m) __attribute__((always_inline)) { if (m == 0) x_load(mt_next, NB); epi(acc[(NB - 1) & 1], NB - 1, m); }); \
    else x_load(mt_next, NB)
    TK_EARLY(1); TK_EARLY(2); TK_EARLY(3); TK_EARLY(4); TK_EARLY(5); TK_EARLY(6); TK_EARLY(7);
#undef TK_EARLY
#pragma unroll 1
    for (int nb = 8; nb < nblk; nb += 2) {
      slab(acc[0], [&](int m) __attribute__((always_inline)) { epi(acc[1], nb - 1, m); });
      if (nb + 1 < nblk) slab(acc[1], [&](int m) __attribute__((always_inline)) { epi(acc[0], nb, m); });
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if ((nblk - 1) & 1) {
#pragma unroll
      for (int st = 0; st < 8; ++st) epi(acc[1], nblk - 1, st);
    } else {
#pragma unroll
      for (int st = 0; st < 8; ++st) epi(acc[0], nblk - 1, st);
    }
    (void)no_side;
  }
#undef TK_PIECE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no LDS-DMA may outlive the block

#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
  record_amax_block(a.amax_out, amax, reinterpret_cast<float*>(smem));      // (no LDS-DMA in flight: vmcnt(0) above)
  if (lane == 0) {
    if (a.range_flag && (!(amax * s_in < 60000.f) || (amax > 0.f && amax * s_in < 0.125f))) atomicMax(a.range_flag, a.site + 1);
  }
}

// ---- d(ln1) = d(qkv) Wqkv^T (K = 768) with LayerNorm-1 backward in the epilogue -----------------------------------------------
// Same ring, same slab; the 768-deep product runs as three operand chunks of 256 columns: chunk c's planes sit in XB while its
// eight slabs (one per 32 output features) accumulate into acc2[0..7].  The next chunk's columns cannot wait in registers
// (accumulators 128 + operand planes 128 + fragment ring 64 + a raw chunk 128 spills ~200 values): they are TOUCHED during
// the current chunk's slabs (one dword per 128-byte line, four per lane) so that the chunk's own loads hit the L2.  The weight planes [row / 32][k / 16][plane] hold a feature block's three
// chunks 32 KB apart, so slab (c, nb) starts at nb * 96 KB + c * 32 KB.  Epilogue = ffx.hip's backward epilogue: lane (r, h)
// holds features 32 nb + 8 q + 4 h + i of token r, LayerNorm statistics need one shuffle per row sum.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void tklb_kernel(TklbArgs a, int n_mt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int n_my = (n_mt - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

  const float s_in = scale_of(a.amax_in);
  const float os = a.wsi / s_in;
  float amax = 0.f;

  const char* wsrc = reinterpret_cast<const char*>(a.W) + wave * 8192 + lane * 16;
  int is_q = 0, is_g = 0;                                   // is_q: slab of the tile, 8 c + nb
  const char* cur_src = wsrc; unsigned cur_dst = 0;
  auto dma_begin = [&]() __attribute__((always_inline)) {
    asm volatile("" : "+s"(is_q), "+s"(is_g));              // (opaque: with the tile's 24 slabs unrolled hipcc otherwise precomputes -- and spills -- every slab's addresses)
    cur_src = wsrc + (long)(is_q & 7) * (3 * TK_SLAB) + (long)(is_q >> 3) * TK_SLAB;
    cur_dst = (unsigned)(uintptr_t)(smem + (is_g & (TK_R - 1)) * TK_SLAB + wave * 8192);
    is_q = is_q + 1 == 24 ? 0 : is_q + 1;
    ++is_g;
  };
#define TK_PIECE(C) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2" \
                                 :: "v"(cur_src + ((C) >> 2) * 4096), "s"(cur_dst + ((C) >> 2) * 4096), "n"(((C) & 3) * 1024) : "memory", "m0")
  auto dma_piece = [&](int c) __attribute__((always_inline)) {
    switch (c) { case 0: TK_PIECE(0); break; case 1: TK_PIECE(1); break; case 2: TK_PIECE(2); break; case 3: TK_PIECE(3); break;
                 case 4: TK_PIECE(4); break; case 5: TK_PIECE(5); break; case 6: TK_PIECE(6); break; default: TK_PIECE(7); break; }
  };
  auto issue_slab = [&]() __attribute__((always_inline)) {
    dma_begin();
#pragma unroll
    for (int c = 0; c < 8; ++c) dma_piece(c);
  };

  u32x4 F[4][4];
  int g = 0;
  const char* rd = smem + lane * 16;
  u32x4 XB[16][2];
  f32x16 acc2[8];
  f32x4 pf[32];                                             // the next operand chunk, raw, fetched one eighth per slab
  float touch = 0.f, tch[4] = {0.f, 0.f, 0.f, 0.f};         // sink of the L2 touches

  auto slab = [&](f32x16& ac, auto side) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    dma_begin();
    asm volatile("" : "+s"(g));
    const int slot = g & (TK_R - 1), nslot = (g + 1) & (TK_R - 1);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      u32x4 (&FB)[4] = F[m & 3];
      u32x4 (&FN)[4] = F[(m + 2) & 3];
      const char* np = rd + (m < 6 ? slot * TK_SLAB + (m + 2) * 4096 : nslot * TK_SLAB + (m - 6) * 4096);
      const int s = 2 * m;
      __builtin_amdgcn_sched_barrier(0);
      ac = mfma16(FB[1], XB[s][0], ac);
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

  // one dword of line j (of 4) of this lane's half row of columns [256 c, 256 c + 256) of tile mt: brings the line into the L2
  auto x_touch = [&](int mt, int c, int j) __attribute__((always_inline)) {
    asm volatile("" : "+s"(mt));
    long tok = (long)mt * 128 + wave * 32 + r;
    tok = tok < a.M ? tok : a.M - 1;
    tch[j] = a.X[tok * 768 + 256 * c + 128 * h + 32 * j];   // (consumed after the chunk's slabs: no wait inside them)
  };

  reinterpret_cast<float*>(smem + TK_LN)[tid] = a.ln_g[tid];
  issue_slab(); issue_slab(); issue_slab();
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) { F[0][i] = *reinterpret_cast<const u32x4*>(rd + i * 1024); F[1][i] = *reinterpret_cast<const u32x4*>(rd + 4096 + i * 1024); }
  const float* lng = reinterpret_cast<const float*>(smem + TK_LN);

  for (int ti = 0; ti < n_my; ++ti) {
    const int mt = (int)blockIdx.x + ti * (int)gridDim.x;
    const int mt_next = ti + 1 < n_my ? mt + (int)gridDim.x : mt;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc2[i][e] = 0.f;
    // chunk c + 1's columns are fetched into pf during chunk c's slabs (chunks 1 and 2); chunk 0 of a tile is loaded at the tile
    // switch (its lines were touched into the L2 during the previous tile's last slabs)
    auto pf_x = [&](int c, int j) __attribute__((always_inline)) {        // columns [256 c, 256 c + 256) of this tile, eighth j
      int mt_o = mt; asm volatile("" : "+s"(mt_o));                       // (addresses formed at use, not carried across the slabs)
      long tok = (long)mt_o * 128 + wave * 32 + r;
      tok = tok < a.M ? tok : a.M - 1;                                    // (rows past M recompute row M - 1)
      const float* xrow = a.X + tok * 768 + 256 * c + 8 * h;              // lane (r, h) holds k = 16 s + 8 h + i of token r
#pragma unroll
      for (int s2 = 2 * j; s2 < 2 * j + 2; ++s2) {
        pf[2 * s2] = *reinterpret_cast<const f32x4*>(xrow + 16 * s2);
        pf[2 * s2 + 1] = *reinterpret_cast<const f32x4*>(xrow + 16 * s2 + 4);
      }
    };
    auto convert = [&]() __attribute__((always_inline)) {                 // pf (raw chunk) -> XB planes
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {
        amax_pin(amax, pf[2 * s2][0], pf[2 * s2][1]); amax_pin(amax, pf[2 * s2][2], pf[2 * s2][3]);
        amax_pin(amax, pf[2 * s2 + 1][0], pf[2 * s2 + 1][1]); amax_pin(amax, pf[2 * s2 + 1][2], pf[2 * s2 + 1][3]);
        split8(pf[2 * s2] * s_in, pf[2 * s2 + 1] * s_in, XB[s2][0], XB[s2][1]);
        if ((s2 & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    };
#pragma unroll
    for (int j = 0; j < 8; ++j) pf_x(0, j);
    convert();
#define TB_SLAB(NB, WORK) slab(acc2[NB], [&](int m) __attribute__((always_inline)) { if (m == 0) { WORK; } })
    TB_SLAB(0, pf_x(1, 0)); TB_SLAB(1, pf_x(1, 1)); TB_SLAB(2, pf_x(1, 2)); TB_SLAB(3, pf_x(1, 3));
    TB_SLAB(4, pf_x(1, 4)); TB_SLAB(5, pf_x(1, 5)); TB_SLAB(6, pf_x(1, 6)); TB_SLAB(7, pf_x(1, 7));
    convert();
    TB_SLAB(0, pf_x(2, 0)); TB_SLAB(1, pf_x(2, 1)); TB_SLAB(2, pf_x(2, 2)); TB_SLAB(3, pf_x(2, 3));
    TB_SLAB(4, pf_x(2, 4)); TB_SLAB(5, pf_x(2, 5)); TB_SLAB(6, pf_x(2, 6)); TB_SLAB(7, pf_x(2, 7));
    convert();
    // (the last chunk prefetches nothing into registers: z AND the bypassing gradient of the epilogue would need 256 arch VGPRs
    // beside the accumulators' copies -- hipcc spills every arriving quad; it touches the next tile's first chunk into the L2)
    TB_SLAB(0, (void)0); TB_SLAB(1, (void)0); TB_SLAB(2, (void)0); TB_SLAB(3, x_touch(mt_next, 0, 0));
    TB_SLAB(4, x_touch(mt_next, 0, 1)); TB_SLAB(5, x_touch(mt_next, 0, 2)); TB_SLAB(6, x_touch(mt_next, 0, 3)); TB_SLAB(7, (void)0);
#undef TB_SLAB
    touch += (tch[0] + tch[1]) + (tch[2] + tch[3]);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

    // ---- epilogue: out = add + LNbwd(d(ln1); z, gamma)  (rowops.hip ln_bwd_kernel; ffx.hip's backward epilogue) ------------------
    int mt_e = mt;
    asm volatile("" : "+s"(mt_e));                          // (row addresses recomputed here, not carried across the slabs)
    long tok_e = (long)mt_e * 128 + wave * 32 + r;
    tok_e = tok_e < a.M ? tok_e : a.M - 1;                  // rows past M recompute and rewrite row M - 1 (unconditional stores)
    const float* zrow = a.Z + tok_e * 256 + 4 * h;
    const float* drow = a.add + tok_e * 256 + 4 * h;
    float* orow = a.Y + tok_e * 256 + 4 * h;
    {
      f32x4 xz[32], ad[32];
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 32; ++i) xz[i] = *reinterpret_cast<const f32x4*>(zrow + 8 * i);
#pragma unroll
      for (int i = 0; i < 32; ++i) ad[i] = *reinterpret_cast<const f32x4*>(drow + 8 * i);
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
      for (int i = 0; i < 32; ++i) {                        // i = 4 nb + q
        const f32x4 gm = *reinterpret_cast<const f32x4*>(lng + 8 * i + 4 * h);
        f32x4 gq = quad(acc2[i >> 2], i & 3) * os * gm;
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
        for (int e = 0; e < 4; ++e) o[e] = (acc2[i >> 2][4 * (i & 3) + e] - m1 - xz[i][e] * m2) * rstd + ad[i][e];
        *reinterpret_cast<f32x4*>(orow + 8 * i) = o;
      }
    }
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");        // (bounds the operations in flight before the next tile, as ffx.hip)
  }
#undef TK_PIECE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
  record_amax_block(a.amax_out, amax, reinterpret_cast<float*>(smem));      // (no LDS-DMA in flight: vmcnt(0) above)
  if (lane == 0) {
    if (a.range_flag && (!(amax * s_in < 60000.f) || (amax > 0.f && amax * s_in < 0.125f))) atomicMax(a.range_flag, a.site + 1);
  }
  if (touch == 1.2345e-30f && a.amax_out) a.amax_out[0] = touch;      // (keeps the touches alive; never true in practice)
}

int launch_tklb(const TklbArgs& a, hipStream_t s) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  RAMP_REQUIRE(a.M > 0 && a.X && a.Z && a.add && a.Y && a.W && a.ln_g, "tklb: null operand");
  RAMP_REQUIRE(al16(a.X) && al16(a.Z) && al16(a.add) && al16(a.Y) && al16(a.W), "tklb: operands must be 16-byte aligned");
  {   // rows past M are recomputed and rewritten from the inputs (unconditional stores): the output may alias none of them
    const size_t yb = (size_t)a.M * 256 * 4;
    RAMP_REQUIRE(!ranges_overlap(a.Y, yb, a.X, (size_t)a.M * 768 * 4) && !ranges_overlap(a.Y, yb, a.Z, yb) && !ranges_overlap(a.Y, yb, a.add, yb),
                 "tklb: the output must not overlap d(qkv), z or the bypassing gradient (no in-place use)");
  }
  const int n_mt = (a.M + 127) / 128;
  hipLaunchKernelGGL(tklb_kernel, dim3(std::min(n_mt, device_cu_count())), dim3(256), TK_LDS, s, a, n_mt);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_tkl(const TklArgs& a, hipStream_t s) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  RAMP_REQUIRE(a.M > 0 && a.N >= 32 && a.N <= 768 && a.N % 32 == 0 && a.X && a.Y && a.W, "tkl: bad operand");
  RAMP_REQUIRE(al16(a.X) && al16(a.Y) && al16(a.W) && al16(a.resid) && a.ldy % 4 == 0 && a.ldr % 4 == 0, "tkl: operands must be 16-byte aligned");
  RAMP_REQUIRE(!a.rowbias || (a.rowvar && a.N == 256 && a.n_var >= 1 && a.n_var <= 4 && a.L >= 1 && a.resid), "tkl: row-variant bias needs N = 256, <= 4 variants, a residual");
  RAMP_REQUIRE(!a.ln_g == !a.ln_b, "tkl: LayerNorm needs gamma and beta");
  {   // rows past M are recomputed and rewritten from the inputs (unconditional stores): an in-place residual (Y == resid) would add
      // the last row's update twice when M is not a multiple of 128
    const size_t yb = ((size_t)(a.M - 1) * a.ldy + a.N) * 4;
    RAMP_REQUIRE(!ranges_overlap(a.Y, yb, a.X, (size_t)a.M * 256 * 4) && !ranges_overlap(a.Y, yb, a.resid, a.resid ? ((size_t)(a.M - 1) * a.ldr + a.N) * 4 : 0),
                 "tkl: the output must not overlap the operand or the residual (no in-place use)");
  }
  const int n_mt = (a.M + 127) / 128;
  const int nb = std::min(n_mt, device_cu_count());          // one 4-wave block per CU
  const bool ln = a.ln_g != nullptr;
  const int epi = (a.resid ? 1 : 0) | (a.rowbias ? 2 : 0);
#define TK_GO(LNV, E, A) hipLaunchKernelGGL((tkl_kernel<LNV, E, A>), dim3(nb), dim3(256), TK_LDS, s, a, n_mt)
#define TK_ABL(A) else if (a.ablate == A && ln && epi == 0) TK_GO(true, 0, A); else if (a.ablate == A && !ln && epi == 1) TK_GO(false, 1, A); \
                  else if (a.ablate == A && !ln && epi == 0) TK_GO(false, 0, A);
  if (a.ablate == 0) {
    if (ln && epi == 0) TK_GO(true, 0, 0);
    else if (!ln && epi == 0) TK_GO(false, 0, 0);
    else if (!ln && epi == 1) TK_GO(false, 1, 0);
    else if (!ln && epi == 3) TK_GO(false, 3, 0);
    else RAMP_REQUIRE(false, "tkl: variant not built");
  }
  TK_ABL(1) TK_ABL(2) TK_ABL(8) TK_ABL(3)
  else RAMP_REQUIRE(false, "tkl: ablation variant not built");
#undef TK_ABL
#undef TK_GO
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

int init_tkl_attributes() {
#define TK_ATTR(LNV, E, A) RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tkl_kernel<LNV, E, A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TK_LDS))
  TK_ATTR(true, 0, 0); TK_ATTR(false, 0, 0); TK_ATTR(false, 1, 0); TK_ATTR(false, 3, 0);
#define TK_ATTR3(A) TK_ATTR(true, 0, A); TK_ATTR(false, 1, A); TK_ATTR(false, 0, A)
  TK_ATTR3(1); TK_ATTR3(2); TK_ATTR3(8); TK_ATTR3(3);
#undef TK_ATTR3
#undef TK_ATTR
  RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tklb_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TK_LDS));
  return 0;
}

}  // namespace ramp

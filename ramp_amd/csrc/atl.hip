// Backward of the transformer block's self-attention half as ONE launch of sample-owning waves:
//
//   dz = dz1 + LayerNorm1-backward( d(qkv) Wqkv^T ; z, gamma ),   d(qkv) = attention-backward(q, k, v, d(o))
//
// (CrossAttention.forward and BasicTransformerBlock.forward differentiated, layers_attention_mini.py:101-127, 132).  It replaces
// atb_kernel / attn2_bwd_kernel + tklb_kernel (tkl.hip): d(qkv) -- 3 KB per token written and read again -- never reaches HBM.
//
// Dataflow = atb_kernel's (atk.hip) with ato_kernel's projection behind it.  A wave owns T = 48 (32) tokens = whole samples; per head it
// computes P^T, dS^T and from them the three gradient tiles dV^T, dQ^T, dK^T [feature 4 g + i][token c] in accumulator registers; each
// such tile IS the B operand of the projection d(ln1)^T[n][token] += W[n][k] d(qkv)^T[k][token], whose 256 x T accumulators stay in the
// accumulation half of the register file for the whole tile (4 heads x 3 parts x 64 = the 768 k of the product).  The weight planes
// (768 KB per tile) stream through a 3-slot LDS ring of 16 KB slabs in the order the gradient tiles are produced (abl_pack); q, k, v,
// d(o) rows reach the wave through two wave-private LDS regions by LDS-DMA, so nothing waits in registers.  d(qkv) is the operand of
// the d(ln1) call site: delayed power-of-two scale, recorded maximum, range guard like every other fp16x3 GEMM.  The LayerNorm backward
// runs on the accumulators in the epilogue, one 16-token group at a time (its z rows stay in registers between the row sums and the outputs).
// The points / waits of a head are described by abl_make_sched below: keep it in step with the kernel's issue order when either changes.
// Round 6 (PK): k is NOT fetched a second time for dQ -- its fp16 planes from the S^T phase (48 registers at T = 48) wait in the accumulation half of
// the register file, where MFMA A operands may live (12 of a head's 72 one-KB operand requests and one operand split less; the 54 accumulation
// registers the kernel did not use made room).  Region A then carries q -> d(o) -> next q.  PK = false is the round-4 kernel (A/B twin).
#include "args_attention.h"
#include "tokmma.h"
#include "atkmma.h"

#include <algorithm>

namespace ramp {

namespace {

constexpr int AL_SLAB = 16 * 1024;                      // 8 output blocks of 16 x one k32 step x 2 planes x 1 KB
constexpr int AL_R = 3;                                 // ring slots: slab g + 2 is requested during slab g
constexpr int AL_NS = 48;                               // slabs per tile: head 4 x part 3 x k32 step 2 x output half 2
constexpr int AL_GAM = AL_R * AL_SLAB;                  // LayerNorm gamma (256 floats)
constexpr int AL_SEL = AL_GAM + 1024;                   // the two selection operands: [b 2][lane 64][16 bytes]
constexpr int AL_RA = AL_SEL + 2048;                    // per wave: operand region A (q -> d(o) -> k -> next q)
constexpr int AL_RB = AL_RA + 4 * AT_VW;                // per wave: operand region B (k -> v -> q -> next k)
constexpr size_t AL_LDS = (size_t)AL_RB + 4 * AT_VW;
static_assert(AL_LDS <= 160 * 1024, "LDS budget");

// ---- the LDS-DMA schedule of a head, simulated at compile time (the kernel's issue order, request by request) -----------------------
// Phases: 0 softmax side (4 NG points: d(o) -> A) | 1 dV (v -> B, then k -> A) | 2 dP / dS (2 NG points: first half of q -> B) | 3 dQ (rest of
// q -> B, then the next head's q -> A) | 4 dK (the next head's k -> B).  A part phase (1, 3, 4) has two k32 steps of 12 points: 6 in the turn +
// contraction, 3 in each of its two slabs (macro-steps 1, 4, 7); every slab also issues the 4 ring pieces of slab g + 2 (macro-steps 0, 2, 4, 6).
// Point p of a phase issues piece p of the phase's stream list, if there is one.  The result: for every consumer the number of requests
// issued BEHIND its data, i.e. the vmcnt it may wait with.
struct AblSched {
  int slab[12];                                         // per slab of the head (-1: requested in the head before, landed at its start)
  int w_do, w_v, w_k, w_q;                              // before the region reads of d(o) (dV), v (dP), k (dQ), q (dK)
  int ops;                                              // operand pieces that found a point (must be all 6 streams x GR)
};
constexpr int abl_phase_pieces(int ph, int GR, bool pk) { return ph == 0 ? GR : ph == 1 ? (pk ? GR : 2 * GR) : ph == 2 ? GR / 2 : ph == 3 ? GR / 2 + GR : GR; }
constexpr AblSched abl_make_sched(int NG, bool pk) {
  const int GR = 4 * NG;
  AblSched r = {};
  int issued = 0, sl = 0;
  int ring_mark[AL_R] = {0, 0, 0};
  int mk_do = 0, mk_v = 0, mk_k = 0, mk_q = 0;
  for (int p = 0; p < 4 * NG; ++p) if (p < abl_phase_pieces(0, GR, pk)) { ++issued; ++r.ops; if (p == GR - 1) mk_do = issued; }
  for (int ph = 1; ph <= 4; ++ph) {
    if (ph == 1) r.w_do = issued - mk_do;
    if (ph == 2) {
      r.w_v = issued - mk_v;
      for (int p = 0; p < 2 * NG; ++p) if (p < abl_phase_pieces(2, GR, pk)) { ++issued; ++r.ops; }
      continue;
    }
    if (ph == 3) r.w_k = issued - mk_k;
    if (ph == 4) r.w_q = issued - mk_q;
    const int np = abl_phase_pieces(ph, GR, pk);
    int p = 0;
    auto pt = [&]() {
      if (p < np) {
        ++issued; ++r.ops;
        if (ph == 1 && p == GR - 1) mk_v = issued;
        if (ph == 1 && p == 2 * GR - 1) mk_k = issued;
        if (ph == 3 && p == GR / 2 - 1) mk_q = issued;
      }
      ++p;
    };
    for (int j = 0; j < 2; ++j) {
      for (int k = 0; k < 6; ++k) pt();
      for (int half = 0; half < 2; ++half) {
        r.slab[sl] = sl >= 2 ? issued - ring_mark[sl % AL_R] : -1;
        for (int m = 0; m < 8; ++m) {
          if (!(m & 1)) { ++issued; if (m == 6) ring_mark[(sl + 2) % AL_R] = issued; }
          if (m == 1 || m == 4 || m == 7) pt();
        }
        ++sl;
      }
    }
  }
  return r;
}
static_assert(abl_make_sched(3, false).ops == 6 * 12 && abl_make_sched(2, false).ops == 6 * 8 && abl_make_sched(3, true).ops == 5 * 12 && abl_make_sched(2, true).ops == 5 * 8,
              "every operand piece of a head needs a point");

}  // namespace

// ---- weight stream ----------------------------------------------------------------------------------------------------------------------
// W [256 n][768 k] row-major (k = q | k | v features, 4 heads x 64 each) -> out [slab 48][nbl 8][plane 2][lane 64][8 halves]; slab
// ((h * 3 + p) * 2 + j) * 2 + half serves head h, part p in the kernel's order (v, q, k), k32 step j of the head's 64 features and
// output features 128 half + 16 nbl + (lane & 15); k order inside the step = the row order of two stacked accumulator tiles (ato_pack).
__global__ void abl_pack_kernel(const float* __restrict__ W, unsigned short* __restrict__ out, float scale) {
  const int idx = blockIdx.x * 256 + threadIdx.x;            // (slab, nbl, lane): 48 * 8 * 64
  if (idx >= AL_NS * 8 * 64) return;
  const int lane = idx & 63, nbl = (idx >> 6) & 7, sl = idx >> 9;
  const int half = sl & 1, j = (sl >> 1) & 1, hp = sl >> 2, p = hp % 3, h = hp / 3;
  const int part = p == 0 ? 2 : p - 1;                       // v, q, k
  const int n = 128 * half + 16 * nbl + (lane & 15), kq = lane >> 4;
  _Float16 hi[8], lo[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int kl = e < 4 ? 4 * kq + e : 16 + 4 * kq + (e - 4);
    const float x = W[(long)n * 768 + 256 * part + 64 * h + 32 * j + kl] * scale;
    hi[e] = (_Float16)x;
    lo[e] = (_Float16)(x - (float)hi[e]);
  }
  unsigned short* o = out + ((long)(sl * 8 + nbl) * 2) * 512 + lane * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) { o[e] = __builtin_bit_cast(unsigned short, hi[e]); o[512 + e] = __builtin_bit_cast(unsigned short, lo[e]); }
}
int abl_pack(const float* W, float scale, unsigned short* out, hipStream_t s) {
  RAMP_REQUIRE(W && out, "abl_pack: null operand");
  hipLaunchKernelGGL(abl_pack_kernel, dim3(AL_NS * 8 * 64 / 256), dim3(256), 0, s, W, out, scale);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

// Order of a head (wave-local): [q (A), k (B) -> planes] S^T, softmax -> P^T | P turned | [d(o) (A) -> planes] dV^T and its 4 slabs | [v (B)]
// dP^T, dS^T | [k (A)] dQ^T and its 4 slabs | dS turned, [q (B)] dK^T and its 4 slabs.  dV comes before dS because it needs neither v nor dS:
// that leaves the softmax phase only d(o) to fetch and gives v the whole dV phase to arrive.  Every slab issues the 4 ring pieces of slab
// g + 2; the operand pieces ride on the "points" of all phases (see the schedule in the kernel).
// STAMP (diagnostic twin, ramp_bench_gemm only): per-wave s_memtime sums of the phases
template <int NG, bool STAMP = false, bool PK = true>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void abl_kernel(AblArgs a, int n_tiles) {
  constexpr int T = 16 * NG;
  constexpr int GR = T / 4;                                 // 4-token groups = LDS-DMA pieces per operand
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (c, g and what follows from them are re-derived from an opaque copy of the lane index at the top of every head and epilogue: left visible,
  // hipcc hoists some thirty lane-invariant addresses and offsets out of both loops and spills them)
  int c = lane & 15, g = lane >> 4;
  const int n_my = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int n_steps = 4 * n_my;

  // (wave-uniform, said so: a loaded value otherwise occupies a vector register for the whole kernel)
  const float s_in = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, scale_of(a.amax_in))));
  const float os = a.wsi / s_in;
  float amax = 0.f;
  unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
  const unsigned long long t_start = STAMP ? __builtin_amdgcn_s_memtime() : 0, r_start = STAMP ? __builtin_amdgcn_s_memrealtime() : 0;
  auto stamp = [&](int k) __attribute__((always_inline)) {
    if (STAMP) { const unsigned long long t = __builtin_amdgcn_s_memtime(); if (tlast) tk[k] += t - tlast; tlast = t; }
  };

  reinterpret_cast<float*>(smem + AL_GAM)[tid] = a.ln_g[tid];     // (published by the first slab barrier; first read in the first epilogue)

  unsigned kmask[NG];                                       // keys (16 kg + 4 g + i) in the sample of query 16 qg + c: bit 4 kg + i
#pragma unroll
  for (int qg = 0; qg < NG; ++qg) {
    unsigned m = 0;
#pragma unroll
    for (int kg = 0; kg < NG; ++kg)
#pragma unroll
      for (int i = 0; i < 4; ++i) m |= ((16 * kg + 4 * g + i) / a.L == (16 * qg + c) / a.L ? 1u : 0u) << (4 * kg + i);
    kmask[qg] = m;
  }
  // selection operands (atb_kernel): as B operand of a k32 step whose A operand pairs two 16-wide blocks, sel[b] picks block b.  Kept in LDS
  // (a lane's 2 x 16 bytes, read back where an operand is turned): 8 registers less over the whole kernel
  if (wave == 0) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      unsigned w[4];
#pragma unroll
      for (int p2 = 0; p2 < 4; ++p2) {
        unsigned v = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int e = 2 * p2 + q;
          if ((e >> 2) == b && 4 * g + (e & 3) == c) v |= 0x3c00u << (16 * q);
        }
        w[p2] = v;
      }
      *reinterpret_cast<u32x4*>(smem + AL_SEL + b * 1024 + lane * 16) = u32x4{w[0], w[1], w[2], w[3]};
    }
  }
  const char* selp = nullptr;
  const u32x2 z2 = {0u, 0u};

  // ---- weight ring (ato_kernel's): wave w copies bytes [4 w KB, +4 KB) of a 16 KB slab as 4 LDS-DMA pieces of 1 KB
  unsigned lane_w = (unsigned)(wave * 4096 + lane * 16);
  int is_g = 0;
  const char* ring_src = nullptr; unsigned ring_dst = 0;
  auto ring_begin = [&]() __attribute__((always_inline)) {
    ring_src = reinterpret_cast<const char*>(a.W) + (long)(is_g % AL_NS) * AL_SLAB;      // scalar
    ring_dst = (unsigned)(uintptr_t)(smem + (is_g % AL_R) * AL_SLAB + wave * 4096);
    ++is_g;
  };
#define AL_PIECE(C) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" \
                                 :: "v"(lane_w), "s"(ring_src), "s"(ring_dst), "n"((C) * 1024) : "memory", "m0")
  auto ring_piece = [&](int cpc) __attribute__((always_inline)) {
    switch (cpc) { case 0: AL_PIECE(0); break; case 1: AL_PIECE(1); break; case 2: AL_PIECE(2); break; default: AL_PIECE(3); break; }
  };
  int gs = 0;                                               // slabs consumed
  const char* rd = nullptr;

  // ---- operand regions: one LDS-DMA instruction = the 256-byte head slices of 4 consecutive tokens ("group": 1 KB + 64 bytes of padding),
  // lane -> row lane & 3, chunk lane >> 2 ([chunk][row]); read back with ds_read_b128 as T-layout rows (lane (c, g) = features 16 fb + 4 g ..
  // of token 16 t + c): conflict-free (ato_kernel's k region)
  const int m_last = a.M - 1;
  const unsigned ra_dst = (unsigned)(uintptr_t)(smem + AL_RA + wave * AT_VW), rb_dst = (unsigned)(uintptr_t)(smem + AL_RB + wave * AT_VW);
  const char* ra_rd = nullptr;
  const char* rb_rd = nullptr;
  auto derive = [&]() __attribute__((always_inline)) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    c = ln & 15; g = ln >> 4;
    lane_w = (unsigned)(wave * 4096 + ln * 16);
    rd = smem + ln * 16;
    selp = smem + AL_SEL + ln * 16;
    ra_rd = smem + AL_RA + wave * AT_VW + (c >> 2) * AT_VG + 64 * g + 16 * (c & 3);
    rb_rd = smem + AL_RB + wave * AT_VW + (c >> 2) * AT_VG + 64 * g + 16 * (c & 3);
  };
  auto op_piece = [&](unsigned dst, const char* base /*scalar: operand + head offset*/, unsigned stride, int tile, int gr) __attribute__((always_inline)) {
    int tk0 = tile * (4 * T) + wave * T + (c & 3);
    asm volatile("" : "+v"(tk0));
    const unsigned off = (unsigned)min(tk0 + 4 * gr, m_last) * stride + 16u * (unsigned)(4 * g + (c >> 2));
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off), "s"(base), "s"(dst + gr * AT_VG) : "memory", "m0");
  };
  auto read_region = [&](const char* rdp, f32x4 (&raw)[4][NG]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < NG; ++t)
#pragma unroll
      for (int fb = 0; fb < 4; ++fb) raw[fb][t] = *reinterpret_cast<const f32x4*>(rdp + 4 * t * AT_VG + 256 * fb);
  };
  const char* qkvb = reinterpret_cast<const char*>(a.QKV);
  const char* dob = reinterpret_cast<const char*>(a.dO);

  // exact power-of-two scale of an operand's first use in the head and its planes [token group][k32 step = feature-block pair]
  auto finish = [&](const f32x4 (&raw)[4][NG], float& scale, u32x4 (&hi)[NG][2], u32x4 (&lo)[NG][2]) __attribute__((always_inline)) {
    if (scale == 0.f) {
      float mx = 0.f;
#pragma unroll
      for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int t = 0; t < NG; ++t) mx = amax4(raw[fb][t], mx);
      scale = pow2_scale(wave_max(mx), 13);
    }
#pragma unroll
    for (int t = 0; t < NG; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        u32x2 h0, l0, h1, l1;
        split4s(raw[2 * j][t], scale, h0, l0); split4s(raw[2 * j + 1][t], scale, h1, l1);
        hi[t][j] = cat2(h0, h1); lo[t][j] = cat2(l0, l1);
      }
  };
  // prologue: slabs 0, 1; the first head's q -> A, k -> B
  ring_begin(); ring_piece(0); ring_piece(1); ring_piece(2); ring_piece(3);
  ring_begin(); ring_piece(0); ring_piece(1); ring_piece(2); ring_piece(3);
#pragma unroll
  for (int gr = 0; gr < GR; ++gr) { op_piece(ra_dst, qkvb, 3072u, (int)blockIdx.x, gr); op_piece(rb_dst, qkvb + 1024, 3072u, (int)blockIdx.x, gr); }

  f32x4 acc[16][NG];                                        // d(ln1)^T: [feature 16 nb + 4 g + i][token 16 t + c]

#pragma unroll 1
  for (int ti = 0; ti < n_my; ++ti) {
  const int tile = (int)blockIdx.x + ti * (int)gridDim.x;
  const long tok0 = (long)tile * (4 * T) + wave * T;
  const bool full = tok0 + T <= a.M;                        // wave-uniform
#pragma unroll
  for (int nb = 0; nb < 16; ++nb)
#pragma unroll
    for (int t = 0; t < NG; ++t) {
      float z0, z1, z2r, z3;
      asm volatile("v_accvgpr_write_b32 %0, 0\n\tv_accvgpr_write_b32 %1, 0\n\tv_accvgpr_write_b32 %2, 0\n\tv_accvgpr_write_b32 %3, 0" : "=a"(z0), "=a"(z1), "=a"(z2r), "=a"(z3));
      acc[nb][t] = f32x4{z0, z1, z2r, z3};
    }
#pragma unroll 1
  for (int h = 0; h < 4; ++h) {
    const int hs = 4 * ti + h;
    const int hs_n = hs + 1 < n_steps ? hs + 1 : hs;        // (the last head re-requests its own rows: unused)
    const int tile_n = (int)blockIdx.x + (hs_n >> 2) * (int)gridDim.x, h_n = hs_n & 3;
    const int hoff = 256 * h;
    derive();
#pragma unroll
    for (int nb = 0; nb < 16; ++nb)
#pragma unroll
      for (int t = 0; t < NG; ++t) asm volatile("" : "+a"(acc[nb][t]));

    // ---- the head's LDS-DMA schedule (abl_make_sched): the pieces ride on "points" spread over every phase of the head, because a CU sustains
    // ~1 KB per wave and ~360 cycles and a wave that issues faster than that stalls with all of its MFMA work behind it; a consumer waits
    // with vmcnt(requests issued behind its data).  Phase and point index are literals after unrolling.
    constexpr AblSched SC = abl_make_sched(NG, PK);
#define AL_WAIT(N) do { switch (N) { \
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break; \
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;   case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break; \
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;   case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break; \
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;   case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break; \
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;   case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break; \
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break; case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break; \
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break; case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break; \
      case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break; case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break; \
      case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break; case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break; \
      case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break; case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break; \
      case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break; case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break; \
      case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break; case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break; \
      default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break; } } while (0)
    auto point = [&](int ph, int p) __attribute__((always_inline)) {
      __builtin_amdgcn_sched_barrier(0);
      if (ph == 0) { if (p < GR) op_piece(ra_dst, dob + hoff, 1024u, tile, p); }                              // d(o)      -> A
      else if (ph == 1) {
        if (p < GR) op_piece(rb_dst, qkvb + 2048 + hoff, 3072u, tile, p);                                     // v         -> B
        else if (!PK && p < 2 * GR) op_piece(ra_dst, qkvb + 1024 + hoff, 3072u, tile, p - GR);                // k again   -> A (PK: its planes are parked)
      }
      else if (ph == 2) { if (p < GR / 2) op_piece(rb_dst, qkvb + hoff, 3072u, tile, p); }                    // q again   -> B, first half
      else if (ph == 3) {
        if (p < GR / 2) op_piece(rb_dst, qkvb + hoff, 3072u, tile, GR / 2 + p);                               // q again   -> B, the rest
        else if (p < GR / 2 + GR) op_piece(ra_dst, qkvb + 256 * h_n, 3072u, tile_n, p - GR / 2);              // next q    -> A
      }
      else { if (p < GR) op_piece(rb_dst, qkvb + 1024 + 256 * h_n, 3072u, tile_n, p); }                       // next k    -> B
      __builtin_amdgcn_sched_barrier(0);
    };

    // the projection of one k32 step of a gradient part: two slabs (output features [0, 128), [128, 256)); B = the gradient tile's planes.
    // Macro-step m = output block 8 half + m: first MFMA | this macro-step's LDS-DMA pieces | fragment reads of m + 1 | the other MFMAs
    // Round 6: the projection's MFMAs are inline asm with the accumulator tied in the accumulation half ("+a"), so that the file can be compiled with
    // -mllvm -amdgpu-mfma-vgpr-form: every OTHER MFMA of the kernel -- the ~190 small result tiles of a head that vector instructions consume at once --
    // then writes VGPRs directly (the compiler pads those hazards itself) instead of AGPRs that have to be read back register by register: 28 % of the
    // head's vector instructions were v_accvgpr_* (DESIGN section 8-2b).  Hazards of the asm, by the guide's table: an accumulate chain on the same
    // accumulator needs no wait states; the B planes are compiler-written VGPRs -> `s_nop 1` opens every slab; the fragments come from ds_read (the compiler
    // waits for the registers it hands in); the accumulators are read by vector instructions in the epilogue only -> `s_nop 11` + a scheduling fence there.
#define AL_ACC(ACC, WA, WB) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(WA), "v"(WB))
    auto project = [&](const u32x4 (&bh)[NG], const u32x4 (&bl)[NG], int ph, int j) __attribute__((always_inline)) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int sl_i = 4 * (ph == 1 ? 0 : ph == 3 ? 1 : 2) + 2 * j + half;      // slab of the head (12 per head: the ring slot index is head-local)
        if (SC.slab[sl_i] >= 0) AL_WAIT(SC.slab[sl_i]);         // (slabs 0, 1 of a head: requested in the head before, landed at its start)
        __builtin_amdgcn_s_barrier();                       // slab gs is complete in LDS; every wave has left slab gs - 1
        stamp(5);
        ring_begin();                                       // slab gs + 2 goes into the slot of slab gs - 1
        const char* sl = rd + (sl_i % AL_R) * AL_SLAB;
        u32x4 wf[2][2];
        wf[0][0] = *reinterpret_cast<const u32x4*>(sl); wf[0][1] = *reinterpret_cast<const u32x4*>(sl + 1024);
        asm volatile("s_nop 1" ::: "memory");                 // (a just-written B plane -> MFMA operand inside asm)
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const int nb = 8 * half + m;
          const u32x4 wh = wf[m & 1][0], wl = wf[m & 1][1];
          __builtin_amdgcn_sched_barrier(0);
          AL_ACC(acc[nb][0], wh, bl[0]);
          __builtin_amdgcn_sched_barrier(0);
          if (!(m & 1)) ring_piece(m >> 1);
          if (m == 1 || m == 4 || m == 7) point(ph, 12 * j + 6 + 3 * half + m / 3);
          __builtin_amdgcn_sched_barrier(0);
          if (m + 1 < 8) {
            wf[(m + 1) & 1][0] = *reinterpret_cast<const u32x4*>(sl + ((m + 1) * 2) * 1024);
            wf[(m + 1) & 1][1] = *reinterpret_cast<const u32x4*>(sl + ((m + 1) * 2 + 1) * 1024);
          }
          AL_ACC(acc[nb][0], wl, bh[0]);
          AL_ACC(acc[nb][0], wh, bh[0]);
#pragma unroll
          for (int t = 1; t < NG; ++t) {
            AL_ACC(acc[nb][t], wh, bl[t]);
            AL_ACC(acc[nb][t], wl, bh[t]);
            AL_ACC(acc[nb][t], wh, bh[t]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        ++gs;
        stamp(6);
      }
    };
    // one k32 step (feature blocks 2 j, 2 j + 1) of a gradient part: the T-layout operand turned (feature on the lane, tokens in the registers;
    // atb_kernel), out^T[d][token] = sum over tokens' (A = turned operand) x (B = planes with the contracted token in the registers, [pair][free
    // token group]); true scale -> recorded maximum -> planes at the call site's scale -> projection.  6 points.
    auto part_step = [&](int ph, const u32x4 (&xh)[NG][2], const u32x4 (&xl)[NG][2], const u32x4 (&bh)[2][NG], const u32x4 (&bl)[2][NG], int j,
                         float oscale) __attribute__((always_inline)) {
      // software-pipelined by hand: the MFMAs of unit u + 1 are issued before the vector work on unit u's result (split into planes, maximum),
      // so that the two pipes overlap -- written one after the other, hipcc runs every unit MFMA -> wait -> split -> MFMA on one accumulator
      const u32x4 sel0 = *reinterpret_cast<const u32x4*>(selp), sel1 = *reinterpret_cast<const u32x4*>(selp + 1024);
      auto tmm = [&](int u) __attribute__((always_inline)) -> f32x4 {      // unit u = f NG + t of the turn
        const int f = u / NG, t = u % NG;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        x = mm32(xh[t][j], f ? sel1 : sel0, x);
        x = mm32(xl[t][j], f ? sel1 : sel0, x);
        return x;
      };
      u32x2 th[2][NG], tl[2][NG];
      {
        f32x4 xc = tmm(0);
#pragma unroll
        for (int u = 0; u < 2 * NG; ++u) {
          f32x4 xn = xc;
          if (u + 1 < 2 * NG) xn = tmm(u + 1);
          unsigned h0, h1, l0, l1;
          split4m(xc, h0, h1, l0, l1);
          th[u / NG][u % NG] = u32x2{h0, h1}; tl[u / NG][u % NG] = u32x2{l0, l1};
          if ((u + 1) % NG == 0) point(ph, 12 * j + u / NG);
          xc = xn;
        }
      }
      planes_fence();                                       // (the turned operand's low planes come out of inline asm: atkmma.h)
      u32x4 ah[2][2], al[2][2];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        ah[f][0] = cat2(th[f][0], th[f][1]); al[f][0] = cat2(tl[f][0], tl[f][1]);
        ah[f][1] = NG == 3 ? cat2(th[f][NG - 1], z2) : u32x4{0u, 0u, 0u, 0u};
        al[f][1] = NG == 3 ? cat2(tl[f][NG - 1], z2) : u32x4{0u, 0u, 0u, 0u};
      }
      auto cmm = [&](int t, f32x4 (&o2)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int pr = 0; pr < (NG == 3 ? 2 : 1); ++pr) {
            o = mm32(ah[f][pr], bl[pr][t], o);
            o = mm32(al[f][pr], bh[pr][t], o);
            o = mm32(ah[f][pr], bh[pr][t], o);
          }
          o2[f] = o;
        }
      };
      u32x4 gh[NG], gl[NG];
      float omax = 0.f;
      {
        f32x4 oc[2];
        cmm(0, oc);
#pragma unroll
        for (int t = 0; t < NG; ++t) {
          f32x4 on[2] = {oc[0], oc[1]};
          if (t + 1 < NG) cmm(t + 1, on);
          const float live = (full || tok0 + 16 * t + c < a.M) ? 1.f : 0.f;      // (tokens past M: not in the recorded maximum; no branch)
          omax = fmaxf(omax, live * amax4(oc[1], amax4(oc[0], 0.f)));
          u32x2 h0, l0, h1, l1;
          split4s(oc[0], oscale * s_in, h0, l0); split4s(oc[1], oscale * s_in, h1, l1);      // (powers of two: one exact scaling)
          gh[t] = cat2(h0, h1); gl[t] = cat2(l0, l1);
          point(ph, 12 * j + 2 + t);
          oc[0] = on[0]; oc[1] = on[1];
        }
      }
      amax = fmaxf(amax, omax * oscale);
      if (NG == 2) point(ph, 12 * j + 4);
      point(ph, 12 * j + 5);
      stamp(4);
      project(gh, gl, ph, j);
    };

    stamp(7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // q in A, k in B (and every older request: the ring slabs of this head's first two)
    stamp(0);
    float sq = 0.f, sk = 0.f, sv = 0.f, sdo = 0.f, sds = 0.f;
    f32x4 pt[NG][NG];                                       // P^T: [key group][query group], keys in the registers, query on the lane
    u32x4 kph[NG][2], kpl[NG][2];                           // PK: k's planes, parked in the accumulation half until dQ
    // ---- S^T = K Q^T -> P^T (softmax over keys, masked to the query's sample); d(o) -> A, one piece at each of its 4 NG points
    {
      u32x4 qh[NG][2], ql[NG][2];
      u32x4 (&kh)[NG][2] = kph; u32x4 (&kl)[NG][2] = kpl;
      { f32x4 raw[4][NG]; read_region(ra_rd, raw); finish(raw, sq, qh, ql); }
      { f32x4 raw[4][NG]; read_region(rb_rd, raw); finish(raw, sk, kh, kl); }
      planes_fence();
      // (both regions are free: their rows are planes)
      const float ssc = 0.125f * 1.4426950408889634f / (sq * sk);
#pragma unroll
      for (int qg = 0; qg < NG; ++qg) {
        f32x4 st[NG];
#pragma unroll
        for (int kg = 0; kg < NG; ++kg) {
          f32x4 x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            x = mm32(kh[kg][j], ql[qg][j], x);
            x = mm32(kl[kg][j], qh[qg][j], x);
            x = mm32(kh[kg][j], qh[qg][j], x);
          }
          st[kg] = x;
        }
        point(0, 4 * qg);
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
        point(0, 4 * qg + 1);
        float sum = 0.f;
#pragma unroll
        for (int kg = 0; kg < NG; ++kg)
#pragma unroll
          for (int i = 0; i < 4; ++i) { const float e = __builtin_amdgcn_exp2f(st[kg][i] - mx); st[kg][i] = e; sum += e; }
        sum = gsum(sum);
        point(0, 4 * qg + 2);
        const float inv = 1.f / sum;
#pragma unroll
        for (int kg = 0; kg < NG; ++kg) pt[kg][qg] = st[kg] * inv;
        point(0, 4 * qg + 3);
      }
    }
    if (PK) {
#pragma unroll
      for (int t = 0; t < NG; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) { asm volatile("" : "+a"(kph[t][j])); asm volatile("" : "+a"(kpl[t][j])); }
    }
    __builtin_amdgcn_sched_barrier(0);
    stamp(1);
    // ---- P with the QUERY in the registers and the key on the lane (turned tile by tile by the selection MFMA): dV's B operand
    u32x4 pqh[2][NG], pql[2][NG];                           // [query-group pair][key group]
    {
      u32x4 pbh[2][NG], pbl[2][NG];
#pragma unroll
      for (int qg = 0; qg < NG; ++qg) {
        u32x2 ph[NG], pl[NG];
#pragma unroll
        for (int kg = 0; kg < NG; ++kg) split4s(pt[kg][qg], 8192.f, ph[kg], pl[kg]);
        pbh[0][qg] = cat2(ph[0], ph[1]); pbl[0][qg] = cat2(pl[0], pl[1]);
        pbh[1][qg] = NG == 3 ? cat2(ph[NG - 1], z2) : u32x4{0u, 0u, 0u, 0u}; pbl[1][qg] = NG == 3 ? cat2(pl[NG - 1], z2) : u32x4{0u, 0u, 0u, 0u};
      }
      planes_fence();
      u32x2 tph[NG][NG], tpl[NG][NG];                       // [query group][key group]
#pragma unroll
      for (int qg = 0; qg < NG; ++qg)
#pragma unroll
        for (int kg = 0; kg < NG; ++kg) {
          const int pr = kg >> 1, b = kg & 1;
          const u32x4 selb = *reinterpret_cast<const u32x4*>(selp + b * 1024);
          f32x4 x = {0.f, 0.f, 0.f, 0.f};
          x = mm32(pbh[pr][qg], selb, x); x = mm32(pbl[pr][qg], selb, x);
          unsigned h0, h1, l0, l1;
          split4m(x, h0, h1, l0, l1); tph[qg][kg] = u32x2{h0, h1}; tpl[qg][kg] = u32x2{l0, l1};
        }
#pragma unroll
      for (int kg = 0; kg < NG; ++kg) {
        pqh[0][kg] = cat2(tph[0][kg], tph[1][kg]); pql[0][kg] = cat2(tpl[0][kg], tpl[1][kg]);
        pqh[1][kg] = NG == 3 ? cat2(tph[NG - 1][kg], z2) : u32x4{0u, 0u, 0u, 0u}; pql[1][kg] = NG == 3 ? cat2(tpl[NG - 1][kg], z2) : u32x4{0u, 0u, 0u, 0u};
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    AL_WAIT(SC.w_do);                                       // d(o) in A
    stamp(2);
    // ---- dV^T = dO^T P: A = d(o) turned (feature on the lane, queries in the registers), B = P (queries in the registers, key on the lane).
    // It needs neither v nor dS, so it comes first: its phases carry v -> B (for dP) and k -> A (for dQ)
    u32x4 doh[NG][2], dol[NG][2];
    { f32x4 raw[4][NG]; read_region(ra_rd, raw); finish(raw, sdo, doh, dol); }
    planes_fence();
#pragma unroll
    for (int j = 0; j < 2; ++j) part_step(1, doh, dol, pqh, pql, j, 1.f / (sdo * 8192.f));
    __builtin_amdgcn_sched_barrier(0);
    AL_WAIT(SC.w_v);                                        // v in B
    // ---- dP^T = V dO^T ; delta_q = sum_k P dP ; dS^T = P^T (dP^T - delta); the first half of q -> B (for dK) at its points
    f32x4 ds[NG][NG];
    {
      u32x4 vh[NG][2], vl[NG][2];
      { f32x4 raw[4][NG]; read_region(rb_rd, raw); finish(raw, sv, vh, vl); }
      planes_fence();
      const float dsc = 1.f / (sv * sdo);
      float dmax = 0.f;
#pragma unroll
      for (int qg = 0; qg < NG; ++qg) {
        f32x4 dp[NG];
#pragma unroll
        for (int kg = 0; kg < NG; ++kg) {
          f32x4 x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            x = mm32(vh[kg][j], dol[qg][j], x);
            x = mm32(vl[kg][j], doh[qg][j], x);
            x = mm32(vh[kg][j], doh[qg][j], x);
          }
          dp[kg] = x * dsc;
        }
        point(2, 2 * qg);
        float delta = 0.f;
#pragma unroll
        for (int kg = 0; kg < NG; ++kg)
#pragma unroll
          for (int i = 0; i < 4; ++i) delta += pt[kg][qg][i] * dp[kg][i];
        delta = gsum(delta);
#pragma unroll
        for (int kg = 0; kg < NG; ++kg) {
#pragma unroll
          for (int i = 0; i < 4; ++i) ds[kg][qg][i] = pt[kg][qg][i] * (dp[kg][i] - delta);
          dmax = amax4(ds[kg][qg], dmax);
        }
        point(2, 2 * qg + 1);
      }
      sds = pow2_scale(wave_max(dmax), 13);
    }
    __builtin_amdgcn_sched_barrier(0);
    // planes of dS^T (x sds) as the token-contracting B operand of dQ: [key-group pair][query group]
    u32x4 sbh[2][NG], sbl[2][NG];
#pragma unroll
    for (int qg = 0; qg < NG; ++qg) {
      u32x2 sh[NG], sl2[NG];
#pragma unroll
      for (int kg = 0; kg < NG; ++kg) split4s(ds[kg][qg], sds, sh[kg], sl2[kg]);
      sbh[0][qg] = cat2(sh[0], sh[1]); sbl[0][qg] = cat2(sl2[0], sl2[1]);
      sbh[1][qg] = NG == 3 ? cat2(sh[NG - 1], z2) : u32x4{0u, 0u, 0u, 0u}; sbl[1][qg] = NG == 3 ? cat2(sl2[NG - 1], z2) : u32x4{0u, 0u, 0u, 0u};
    }
    planes_fence();
    stamp(3);
    // ---- dQ^T = K^T dS^T / 8: A = k turned, B = dS^T (keys in the registers, query on the lane); its phases carry the rest of q -> B and the
    // NEXT head's q -> A
    if (PK) {
#pragma unroll
      for (int j = 0; j < 2; ++j) part_step(3, kph, kpl, sbh, sbl, j, 0.125f / (sk * sds));
    } else {
      AL_WAIT(SC.w_k);                                      // k in A
      u32x4 kh[NG][2], kl[NG][2];
      { f32x4 raw[4][NG]; read_region(ra_rd, raw); finish(raw, sk, kh, kl); }
      planes_fence();
#pragma unroll
      for (int j = 0; j < 2; ++j) part_step(3, kh, kl, sbh, sbl, j, 0.125f / (sk * sds));
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- dK^T = Q^T dS / 8: dS turned tile by tile (query in the registers, key on the lane); its phases carry the next head's k -> B
    {
      u32x4 sqh[2][NG], sql[2][NG];                         // [query-group pair][key group]
      {
        u32x2 tsh[NG][NG], tsl[NG][NG];
#pragma unroll
        for (int qg = 0; qg < NG; ++qg)
#pragma unroll
          for (int kg = 0; kg < NG; ++kg) {
            const int pr = kg >> 1, b = kg & 1;
            const u32x4 selb = *reinterpret_cast<const u32x4*>(selp + b * 1024);
            f32x4 y = {0.f, 0.f, 0.f, 0.f};
            y = mm32(sbh[pr][qg], selb, y); y = mm32(sbl[pr][qg], selb, y);
            unsigned h0, h1, l0, l1;
            split4m(y, h0, h1, l0, l1); tsh[qg][kg] = u32x2{h0, h1}; tsl[qg][kg] = u32x2{l0, l1};
          }
#pragma unroll
        for (int kg = 0; kg < NG; ++kg) {
          sqh[0][kg] = cat2(tsh[0][kg], tsh[1][kg]); sql[0][kg] = cat2(tsl[0][kg], tsl[1][kg]);
          sqh[1][kg] = NG == 3 ? cat2(tsh[NG - 1][kg], z2) : u32x4{0u, 0u, 0u, 0u}; sql[1][kg] = NG == 3 ? cat2(tsl[NG - 1][kg], z2) : u32x4{0u, 0u, 0u, 0u};
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      AL_WAIT(SC.w_q);                                      // q in B
      u32x4 qh[NG][2], ql[NG][2];
      { f32x4 raw[4][NG]; read_region(rb_rd, raw); finish(raw, sq, qh, ql); }
      planes_fence();
#pragma unroll
      for (int j = 0; j < 2; ++j) part_step(4, qh, ql, sqh, sql, j, 0.125f / (sq * sds));
    }
#undef AL_WAIT
#undef AL_ACC
  }
    // ================= epilogue of the tile: dz = add + LNbwd(d(ln1); z, gamma)  (rowops.hip ln_bwd_kernel; tklb_kernel's epilogue) ==========
    // lane (c, g) holds features 16 nb + 4 g + i of tokens 16 t + c: row sums = in-lane over (nb, i) + two shuffles over g.  One token group
    // at a time, so that its 64 z values per lane stay in registers between the sums and the outputs (z is read once); the next group's z
    // rows are requested before this group's outputs.
    {
      asm volatile("s_nop 11" ::: "memory");                  // (the last asm MFMA's accumulator -> the epilogue's reads: 12 wait states behind an 8-pass MFMA)
      __builtin_amdgcn_sched_barrier(0);
      derive();
      const float* gam = reinterpret_cast<const float*>(smem + AL_GAM) + 4 * g;
      const char* zb = reinterpret_cast<const char*>(a.Z);
      const char* ab = reinterpret_cast<const char*>(a.add);
      char* yb = reinterpret_cast<char*>(a.Y);
      int tk0 = (int)tok0 + c;
      asm volatile("" : "+v"(tk0));
      f32x4 zr[2][16];
      auto z_load = [&](int t) __attribute__((always_inline)) {
        const unsigned off = (unsigned)min(tk0 + 16 * t, m_last) * 1024u + 16u * (unsigned)g;
#pragma unroll
        for (int nb = 0; nb < 16; ++nb) zr[t & 1][nb] = *reinterpret_cast<const f32x4*>(zb + off + 64 * nb);
      };
      z_load(0);
#pragma unroll
      for (int t = 0; t < NG; ++t) {
        const unsigned yoff = (unsigned)min(tk0 + 16 * t, m_last) * 1024u + 16u * (unsigned)g;
        f32x4 ad[2][4];
        auto a_load = [&](int b4) __attribute__((always_inline)) {
#pragma unroll
          for (int q = 0; q < 4; ++q) ad[b4 & 1][q] = *reinterpret_cast<const f32x4*>(ab + yoff + 64 * (4 * b4 + q));
        };
        a_load(0);
        if (t + 1 < NG) z_load(t + 1);
        a_load(1);
        __builtin_amdgcn_sched_barrier(0);
        // the four sums of the row about a pivot (the lane's first element of the row's first feature block, taken from lane group 0)
        const float piv = __shfl(zr[t & 1][0][0], c);
        float s1 = 0.f, s2 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int nb = 0; nb < 16; ++nb) {
          const f32x4 gm = *reinterpret_cast<const f32x4*>(gam + 16 * nb);
          const f32x4 gq = acc[nb][t] * os * gm;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = zr[t & 1][nb][e] - piv;
            zr[t & 1][nb][e] = d;
            s1 += d; s2 += d * d; t1 += gq[e]; t2 += gq[e] * d;
          }
        }
        s1 = gsum(s1);
        s2 = gsum(s2);
        t1 = gsum(t1);
        t2 = gsum(t2);
        const float ms = s1 * (1.f / 256.f);
        const float var = fmaxf(s2 * (1.f / 256.f) - ms * ms, 0.f);
        const float rstd = 1.f / sqrtf(var + 1e-5f);
        const float m1 = t1 * (1.f / 256.f);
        const float m2 = rstd * (t2 - ms * t1) * (1.f / 256.f);     // mean of g x-hat
        const float r2 = rstd * rstd * m2;
        // out = (g - m1 - x-hat m2) rstd + add, x-hat = (d - ms) rstd
        // (every store of a full wave tile unconditional: a store behind a per-lane predicate sits in its own basic block behind vmcnt(0))
        auto pass2 = [&](bool pred) __attribute__((always_inline)) {
#pragma unroll
          for (int b4 = 0; b4 < 4; ++b4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int nb = 4 * b4 + q;
              const f32x4 gm = *reinterpret_cast<const f32x4*>(gam + 16 * nb);
              const f32x4 gq = acc[nb][t] * os * gm;
              f32x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = (gq[e] - m1) * rstd - (zr[t & 1][nb][e] - ms) * r2 + ad[b4 & 1][q][e];
              if (!pred || tk0 + 16 * t < a.M) *reinterpret_cast<f32x4*>(yb + yoff + 64 * nb) = o;
            }
            if (b4 + 2 < 4) a_load(b4 + 2);
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        if (full) pass2(false); else pass2(true);
      }
    }
    stamp(7);
  }
#undef AL_PIECE
  if (STAMP && a.stamps && lane == 0) {
    unsigned long long* o = a.stamps + ((long)blockIdx.x * 4 + wave) * 10;
    for (int k = 0; k < 8; ++k) o[k] = tk[k];
    o[8] = __builtin_amdgcn_s_memtime() - t_start;          // shader cycles of the whole kernel ..
    o[9] = __builtin_amdgcn_s_memrealtime() - r_start;      // .. over 100 MHz ticks: the clock it ran at
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no LDS-DMA may outlive the block

  amax = wave_max(amax);
  record_amax_block_guarded(a.amax_out, amax, reinterpret_cast<float*>(smem), a.range_flag, s_in, a.site);      // (no LDS-DMA in flight: vmcnt(0) above; it starts with a barrier)
}

int launch_abl(const AblArgs& a, hipStream_t s) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  int ng = 0;
  RAMP_REQUIRE(ato_applicable(a.M, a.L, &ng), "abl: tokens per sample must divide 48 or 32 (and M be whole samples)");
  RAMP_REQUIRE(a.QKV && a.dO && a.W && a.Z && a.add && a.ln_g && a.Y, "abl: null operand");
  RAMP_REQUIRE(al16(a.QKV) && al16(a.dO) && al16(a.W) && al16(a.Z) && al16(a.add) && al16(a.Y), "abl: operands must be 16-byte aligned");
  RAMP_REQUIRE((long)a.M * 3072 < (1l << 32), "abl: 32-bit row offsets bound M to 1398100 tokens");
  {
    const size_t yb = (size_t)a.M * 1024;
    RAMP_REQUIRE(!ranges_overlap(a.Y, yb, a.QKV, 3 * yb) && !ranges_overlap(a.Y, yb, a.dO, yb) && !ranges_overlap(a.Y, yb, a.Z, yb) && !ranges_overlap(a.Y, yb, a.add, yb),
                 "abl: the output must not overlap the operands (other blocks still read them)");
  }
  const int T = 16 * ng, n_tiles = (a.M + 4 * T - 1) / (4 * T);
  const int nb = std::min(n_tiles, device_cu_count());
  if (a.stamps) {
    RAMP_REQUIRE(ng == 3, "abl: the stamped twin exists for T = 48");
    hipLaunchKernelGGL((abl_kernel<3, true>), dim3(nb), dim3(256), AL_LDS, s, a, n_tiles);
  }
  else if (a.no_park) {
    if (ng == 3) hipLaunchKernelGGL((abl_kernel<3, false, false>), dim3(nb), dim3(256), AL_LDS, s, a, n_tiles);
    else hipLaunchKernelGGL((abl_kernel<2, false, false>), dim3(nb), dim3(256), AL_LDS, s, a, n_tiles);
  }
  else if (ng == 3) hipLaunchKernelGGL((abl_kernel<3>), dim3(nb), dim3(256), AL_LDS, s, a, n_tiles);
  else hipLaunchKernelGGL((abl_kernel<2>), dim3(nb), dim3(256), AL_LDS, s, a, n_tiles);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

int init_atl_attributes() {
  RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&abl_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AL_LDS));
  RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&abl_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AL_LDS));
  RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&abl_kernel<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AL_LDS));
  RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&abl_kernel<3, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AL_LDS));
  RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&abl_kernel<2, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AL_LDS));
  return 0;
}

}  // namespace ramp

// 4-head x 64 softmax self-attention inside each row of L <= 64 tokens, forward and dX, on the fp32 matrix
// cores (v_mfma_f32_16x16x4_f32: exact fp32).   Reference: CrossAttention.forward used as self-attention
// (layers_attention_mini.py:101-127) and its autograd.
//
// One wave per (row, head) pair, a 256-thread block = the 4 heads of one row.  Everything is computed in the
// TRANSPOSED orientation S^T = K Q^T, so that
//   * the MFMA C-layout puts the query on the lane (col = lane & 15) and the keys in registers / lane groups:
//     the softmax over keys is an in-lane reduction plus two xor-shuffles (16, 32);
//   * P^T (and dS^T) in their accumulator registers ARE the B operand of the next products that contract over
//     the key index (O^T = V^T P^T, dQ^T = K^T dS^T): register `reg` of lane group g is key 16 t + 4 g + reg,
//     which is exactly the k index the A operand is read with — no data movement;
//   * delta_i = sum_j P_ij dP_ij is again in-lane + two shuffles (no O recomputation).
// Products that contract over the query index (dV, dK) need P and dS with the key on the lane: one transpose
// each through LDS.  Row-major operand fragments (K, Q, V, dO for S^T and dP^T) are loaded straight from
// global memory in MFMA layout (64 contiguous bytes per lane, k order permuted identically for A and B);
// transposed operand access (V^T, K^T, dO^T, Q^T) goes through one [Lp][68] LDS tile per wave, reused
// phase after phase, which also hosts the transposes: 13 KB per wave at L = 48 -> 12 waves per CU.
#include "args_attention.h"

namespace ramp {

// Every wave owns one (row, head) problem and its own LDS tile, so nothing is shared across the block: the waves only
// need their own LDS writes ordered before their own (cross-lane) LDS reads.  A wave's LDS instructions execute in
// issue order, so a compiler-level fence is enough; a block barrier here would make four independent problems march
// in lockstep through their load / MFMA / store phases.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int L> struct A2 {
  static constexpr int T = (L + 15) / 16;       // 16-row tiles
  static constexpr int LP = 16 * T;
  static constexpr int LD = 68;                  // LDS tile row stride (floats): 4*68 = 16 mod 32 banks
  static constexpr int TS = LP + 1;              // transpose row stride
  static constexpr int TILE = LP * LD > LP * TS ? LP * LD : LP * TS;   // floats per wave
};

// lane (rr = lane & 15, kq = lane >> 4) holds X[16 t + rr][16 kq + s], s = 0..15 (zero for rows >= L)
template <int L>
__device__ __forceinline__ void load_rowmajor(float (&f)[A2<L>::T][16], const float* __restrict__ base, int ld,
                                              int lane) {
  const int rr = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int t = 0; t < A2<L>::T; ++t) {
    const int row = 16 * t + rr;
    const bool ok = row < L;
    const float* p = base + (long)(ok ? row : 0) * ld + 16 * kq;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = ok ? *reinterpret_cast<const f32x4*>(p + 4 * q) : f32x4{0, 0, 0, 0};
      f[t][4 * q] = v[0]; f[t][4 * q + 1] = v[1]; f[t][4 * q + 2] = v[2]; f[t][4 * q + 3] = v[3];
    }
  }
}

// wave-cooperative copy of X[0..L)[0..64) (row stride ld) into the wave's LDS tile [LP][LD], zero rows >= L
template <int L>
__device__ __forceinline__ void load_tile_lds(float* tile, const float* __restrict__ base, int ld, int lane) {
  using C = A2<L>;
  for (int idx = lane; idx < C::LP * 16; idx += 64) {
    const int row = idx >> 4, q = idx & 15;
    f32x4 v = {0, 0, 0, 0};
    if (row < L) v = *reinterpret_cast<const f32x4*>(base + (long)row * ld + q * 4);
    *reinterpret_cast<f32x4*>(tile + row * C::LD + q * 4) = v;
  }
}

// acc[tj][ti] (C-layout) = sum_d A[16 tj + .][d] * B[16 ti + .][d]  from row-major fragments
template <int L>
__device__ __forceinline__ void mma_rowmajor(f32x4 (&acc)[A2<L>::T][A2<L>::T], const float (&fa)[A2<L>::T][16],
                                             const float (&fb)[A2<L>::T][16]) {
#pragma unroll
  for (int tj = 0; tj < A2<L>::T; ++tj)
#pragma unroll
    for (int ti = 0; ti < A2<L>::T; ++ti) {
      f32x4 c = {0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 16; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[tj][s], fb[ti][s], c, 0, 0, 0);
      acc[tj][ti] = c;
    }
}

// out^T[dd][col] = sum_k X[k][dd] * Bm[k][col]: X from the LDS tile (transposed access), Bm in accumulator
// registers (key/query index k = 16 t + 4 (lane >> 4) + reg).  Writes float4 rows to `dst` (row stride ld):
// dst[16 tc + (lane & 15)][16 dt + 4 (lane >> 4) + 0..3] = scale * result, for rows < L.
template <int L>
__device__ __forceinline__ void mma_transposed_store(const float* tile, const f32x4 (&bm)[A2<L>::T][A2<L>::T],
                                                     float* __restrict__ dst, int ld, float scale, int lane) {
  using C = A2<L>;
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int tc = 0; tc < C::T; ++tc) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 acc = {0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < C::T; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const float av = tile[(16 * t + 4 * g + reg) * C::LD + 16 * dt + c];
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bm[t][tc][reg], acc, 0, 0, 0);
        }
      const int row = 16 * tc + c;
      if (row < L) *reinterpret_cast<f32x4*>(dst + (long)row * ld + 16 * dt + 4 * g) = acc * scale;
    }
  }
}

// P^T = softmax over keys of scale * S^T, in place.  Lane (c, g) holds S^T[16 tj + 4 g + reg][16 ti + c].
template <int L>
__device__ __forceinline__ void softmax_keys(f32x4 (&s)[A2<L>::T][A2<L>::T], int lane) {
  using C = A2<L>;
  const int g = lane >> 4;
#pragma unroll
  for (int ti = 0; ti < C::T; ++ti) {
    float mx = -3.0e38f;
#pragma unroll
    for (int tj = 0; tj < C::T; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const bool ok = 16 * tj + 4 * g + reg < L;
        const float v = ok ? s[tj][ti][reg] * 0.125f : -3.0e38f;
        s[tj][ti][reg] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int tj = 0; tj < C::T; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const bool ok = 16 * tj + 4 * g + reg < L;
        const float e = ok ? expf(s[tj][ti][reg] - mx) : 0.f;
        s[tj][ti][reg] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.f / sum;
#pragma unroll
    for (int tj = 0; tj < C::T; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) s[tj][ti][reg] *= inv;
  }
}

// in-place transpose of a (key-major, query-on-lane) register tile set into (query-major, key-on-lane)
template <int L>
__device__ __forceinline__ void transpose_tiles(f32x4 (&m)[A2<L>::T][A2<L>::T], float* tile, int lane) {
  using C = A2<L>;
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int tj = 0; tj < C::T; ++tj)
#pragma unroll
    for (int ti = 0; ti < C::T; ++ti)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) tile[(16 * tj + 4 * g + reg) * C::TS + 16 * ti + c] = m[tj][ti][reg];
  wave_sync();
  f32x4 n[C::T][C::T];
#pragma unroll
  for (int ti = 0; ti < C::T; ++ti)
#pragma unroll
    for (int tj = 0; tj < C::T; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) n[ti][tj][reg] = tile[(16 * tj + c) * C::TS + 16 * ti + 4 * g + reg];
  wave_sync();
#pragma unroll
  for (int a = 0; a < C::T; ++a)
#pragma unroll
    for (int b = 0; b < C::T; ++b) m[a][b] = n[a][b];
}

template <int L>
__global__ __launch_bounds__(256) void attn2_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ o, int n_pairs) {
  using C = A2<L>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* tile = lds + wave * C::TILE;
  const int pair = blockIdx.x * 4 + wave;
  const bool live = pair < n_pairs;
  const int pr = live ? pair : 0;
  const int row = pr >> 2, head = pr & 3;
  const float* qb = qkv + (long)row * L * 768 + head * 64;

  f32x4 st[C::T][C::T];
  {
    float fk[C::T][16], fq[C::T][16];
    load_rowmajor<L>(fk, qb + 256, 768, lane);
    load_rowmajor<L>(fq, qb, 768, lane);
    mma_rowmajor<L>(st, fk, fq);                       // S^T = K Q^T
  }
  softmax_keys<L>(st, lane);                            // P^T
  load_tile_lds<L>(tile, qb + 512, 768, lane);          // V
  wave_sync();
  float* ob = live ? o + (long)row * L * 256 + head * 64 : nullptr;
  if (live) mma_transposed_store<L>(tile, st, ob, 256, 1.f, lane);    // O^T = V^T P^T
}

template <int L>
__global__ __launch_bounds__(256, (L == 48 ? 3 : L == 24 ? 4 : L == 64 ? 2 : L == 32 ? 3 : 1)) void attn2_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                         float* __restrict__ dqkv, int n_pairs) {
  using C = A2<L>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* tile = lds + wave * C::TILE;
  const int pair = blockIdx.x * 4 + wave;
  const bool live = pair < n_pairs;
  const int pr = live ? pair : 0;
  const int row = pr >> 2, head = pr & 3;
  const float* qb = qkv + (long)row * L * 768 + head * 64;
  const float* db = dout + (long)row * L * 256 + head * 64;
  float* gb = dqkv + (long)row * L * 768 + head * 64;

  f32x4 pt[C::T][C::T], ds[C::T][C::T];
  {
    float fa[C::T][16], fb[C::T][16];
    load_rowmajor<L>(fa, qb + 256, 768, lane);         // K
    load_rowmajor<L>(fb, qb, 768, lane);               // Q
    mma_rowmajor<L>(pt, fa, fb);                       // S^T
    softmax_keys<L>(pt, lane);                         // P^T
    load_rowmajor<L>(fa, qb + 512, 768, lane);         // V
    load_rowmajor<L>(fb, db, 256, lane);               // dO
    mma_rowmajor<L>(ds, fa, fb);                       // dP^T = V dO^T
  }
  // delta_i = sum_j P_ij dP_ij ; dS^T = P^T (dP^T - delta)
#pragma unroll
  for (int ti = 0; ti < C::T; ++ti) {
    float delta = 0.f;
#pragma unroll
    for (int tj = 0; tj < C::T; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) delta += pt[tj][ti][reg] * ds[tj][ti][reg];
    delta += __shfl_xor(delta, 16);
    delta += __shfl_xor(delta, 32);
#pragma unroll
    for (int tj = 0; tj < C::T; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) ds[tj][ti][reg] = pt[tj][ti][reg] * (ds[tj][ti][reg] - delta);
  }
  // dQ^T = K^T dS^T / 8
  load_tile_lds<L>(tile, qb + 256, 768, lane);
  wave_sync();
  if (live) mma_transposed_store<L>(tile, ds, gb, 768, 0.125f, lane);
  wave_sync();
  // key on the lane for the products that contract over the query
  transpose_tiles<L>(pt, tile, lane);                   // P  [query-major]
  transpose_tiles<L>(ds, tile, lane);                   // dS [query-major]
  // dV^T = dO^T P
  load_tile_lds<L>(tile, db, 256, lane);
  wave_sync();
  if (live) mma_transposed_store<L>(tile, pt, gb + 512, 768, 1.f, lane);
  wave_sync();
  // dK^T = Q^T dS / 8
  load_tile_lds<L>(tile, qb, 768, lane);
  wave_sync();
  if (live) mma_transposed_store<L>(tile, ds, gb + 256, 768, 0.125f, lane);
}

template <int L> static int attn2_set_attr() {
  const size_t bytes = 4 * (size_t)A2<L>::TILE * sizeof(float);
  RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2_fwd_kernel<L>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2_bwd_kernel<L>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return 0;
}
// every token count a level of a horizon H = 8, 16, ..., 64 can have (H, H/2, H/4, H/8): the reference takes any
// n_support_points divisible by 8 (UnetInference.py:42-56); the kernels mask rows >= L inside tiles of 16
#define RAMP_ATTN_LENGTHS(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(10) X(12) X(14) X(16) X(20) X(24) X(28) X(32) X(40) X(48) X(56) X(64)
int init_attention_attributes() {
#define RAMP_ATTN_ATTR(LV) if (int e = attn2_set_attr<LV>()) return e;
  RAMP_ATTN_LENGTHS(RAMP_ATTN_ATTR)
#undef RAMP_ATTN_ATTR
  return 0;
}

template <int L> static int fwd_launch(const float* qkv, float* o, int R, hipStream_t s) {
  const int n_pairs = R * 4;
  hipLaunchKernelGGL(attn2_fwd_kernel<L>, dim3(R), dim3(256), 4 * (size_t)A2<L>::TILE * sizeof(float), s, qkv, o, n_pairs);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}
template <int L> static int bwd_launch(const float* qkv, const float* dout, float* dqkv, int R, hipStream_t s) {
  const int n_pairs = R * 4;
  hipLaunchKernelGGL(attn2_bwd_kernel<L>, dim3(R), dim3(256), 4 * (size_t)A2<L>::TILE * sizeof(float), s, qkv, dout,
                     dqkv, n_pairs);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

#define RAMP_ATTN_CASE_F(LV) case LV: return fwd_launch<LV>(qkv, o, R, s);
#define RAMP_ATTN_CASE_B(LV) case LV: return bwd_launch<LV>(qkv, dout, dqkv, R, s);

int launch_attn_fwd(const float* qkv, float* o, int R, int L, hipStream_t s) {
  RAMP_REQUIRE(R > 0, "empty attention");
  switch (L) { RAMP_ATTN_LENGTHS(RAMP_ATTN_CASE_F) default: break; }
  RAMP_REQUIRE(false, "attention kernels cover the level lengths of horizons 8, 16, ..., 64 only");
}
int launch_attn_bwd(const float* qkv, const float* dout, float* dqkv, int R, int L, hipStream_t s) {
  RAMP_REQUIRE(R > 0, "empty attention");
  switch (L) { RAMP_ATTN_LENGTHS(RAMP_ATTN_CASE_B) default: break; }
  RAMP_REQUIRE(false, "attention kernels cover the level lengths of horizons 8, 16, ..., 64 only");
}

}  // namespace ramp

// MFMA shape probe (round 6, review item 1): the inner loop of the token-owning fused feed-forward (ffx.hip) reduced to what
// bounds it -- the fp16x3 three-product MFMA pattern fed by ds_read_b128 fragment reads from a 4 x 32 KB LDS ring, one barrier
// per slab, one wave per SIMD, 128 accumulator registers -- in BOTH MFMA shapes:
//   SHAPE 0: v_mfma_f32_32x32x16_f16, 6 per macro-step (what ffx / tkl ship): fragment = 32 features x 16 k
//   SHAPE 1: v_mfma_f32_16x16x32_f16, 12 per macro-step: fragment = 16 features x 32 k, each feeds the token halves 0-15 / 16-31
// Same LDS bytes per FLOP (4 fragments of 1 KB per macro-step), same B-operand registers (32 tokens x 256 k as two planes = 128),
// same accumulator count.  Options: LDS-DMA refill of the ring from a 3 MB L2-resident stream (one 1 KB piece per wave and
// macro-step, as ffx), V pinned vector instructions per macro-step (v_fma_f32 on private registers: ffx forward carries ~18, the
// backward ~10), data random or zero.  Every wave stamps s_memtime / s_memrealtime around its loop: in-kernel clock =
// d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back (6)).  Diagnostic program, not part of the library.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o ramp_amd/lib/mfma_shape_probe ramp_amd/tools/mfma_shape_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int SLAB = 32 * 1024, RING = 4;

__device__ __forceinline__ f32x16 mma32(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
// (as inline asm with the accumulator tied in the accumulation half of the file: through the builtin hipcc permutes the 32 loop-carried
// quads through VGPRs on every trip -- 150 register moves per 288 MFMAs.  Back-to-back MFMAs on the same accumulator need no wait states.)
__device__ __forceinline__ f32x4 mma16(u32x4 a, u32x4 b, f32x4 c) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  return c;
}

template <int I, int N, int V>
__device__ __forceinline__ void sched_seq() {
  if constexpr (I < N) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    constexpr int RD = N == 6 ? ((I == 0 || I == 2) ? 2 : 0) : ((I == 0 || I == 2 || I == 4 || I == 6) ? 1 : 0);
    if constexpr (RD > 0) __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
    constexpr int NV = (I + 1) * V / N - I * V / N;
    if constexpr (NV > 0) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
    sched_seq<I + 1, N, V>();
  }
}

// one launch: every block runs `n_slabs` slabs of 8 macro-steps
template <int SHAPE, int V, bool DMA, int DPOS = -1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void probe_kernel(const unsigned short* __restrict__ wstream, const unsigned short* __restrict__ xb, float* __restrict__ out,
                  unsigned long long* __restrict__ stamps, int n_slabs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ring <- the first 128 KB of the stream (random fp16 planes), every wave a quarter of every slot
  for (int i = tid; i < RING * SLAB / 16; i += 256) reinterpret_cast<u32x4*>(smem)[i] = reinterpret_cast<const u32x4*>(wstream)[i];
  u32x4 XB[16][2];                                           // B operand: 32 tokens x 256 k as two planes (128 registers)
#pragma unroll
  for (int s = 0; s < 16; ++s) { XB[s][0] = reinterpret_cast<const u32x4*>(xb)[(wave * 32 + 2 * s) * 64 + lane]; XB[s][1] = reinterpret_cast<const u32x4*>(xb)[(wave * 32 + 2 * s + 1) * 64 + lane]; }
  f32x16 acc[SHAPE == 0 ? 8 : 1];                          // 128 accumulator registers either way
  f32x4 a4[SHAPE == 0 ? 1 : 32];
#pragma unroll
  for (int i = 0; i < (SHAPE == 0 ? 8 : 1); ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
  for (int i = 0; i < (SHAPE == 0 ? 1 : 32); ++i) a4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float fv[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};    // the filler's private chains
  const float fm = 0.999f + 1e-9f * lane;
  __syncthreads();

  const char* wsrc = reinterpret_cast<const char*>(wstream) + wave * 8192 + lane * 16;
  const char* rd = smem + lane * 16;
  u32x4 F[3][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { F[0][i] = *reinterpret_cast<const u32x4*>(rd + i * 1024); F[1][i] = *reinterpret_cast<const u32x4*>(rd + 4096 + i * 1024); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  int q = 0;                                                 // position in the 96-slab stream
  for (int g = 0; g < n_slabs; g += 3) {                     // three slabs per trip: the fragment ring advances 8 = 2 mod 3 per slab
#pragma unroll
    for (int sub = 0; sub < 3; ++sub) {
      const int FO = (2 * sub) % 3;
      const int slot = (g + sub) & (RING - 1), nslot = (g + sub + 1) & (RING - 1);
      if (DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const char* cur_src = wsrc + (long)q * SLAB;
      const unsigned cur_dst = (unsigned)(uintptr_t)(smem + ((g + sub + 3) & (RING - 1)) * SLAB + wave * 8192);
      q = q + 1 == 96 ? 0 : q + 1;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        u32x4 (&FB)[4] = F[(m + FO) % 3];
        u32x4 (&FN)[4] = F[(m + 2 + FO) % 3];
        const char* np = rd + (m < 6 ? slot * SLAB + (m + 2) * 4096 : nslot * SLAB + (m - 6) * 4096);
        __builtin_amdgcn_sched_barrier(0);
        auto dma_now = [&]() __attribute__((always_inline)) {
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2"
                       :: "v"(cur_src + (m >> 2) * 4096), "s"(cur_dst + (m >> 2) * 4096), "n"((m & 3) * 1024) : "memory", "m0");
        };
        if (DMA && DPOS < 0) dma_now();                      // (ffx's place: ahead of the macro-step's reads; DPOS >= 0: behind MFMA DPOS instead)
        __builtin_amdgcn_sched_barrier(0);
        const int s = (2 * m + sub) & 15;
        // issue order written out and pinned gap by gap (sched_barrier(0) after every MFMA and its gap's share of the side work): the
        // fragment reads behind the first MFMAs, at most two per 32-cycle / one per 16-cycle gap, the fillers spread evenly
        int rd_i = 0, fv_i = 0;
        auto gap = [&](int i, int n) __attribute__((always_inline)) {
          const int nr = n == 6 ? ((i == 0 || i == 2) ? 2 : 0) : ((i == 0 || i == 2 || i == 4 || i == 6) ? 1 : 0);
          for (int k = 0; k < nr; ++k, ++rd_i) FN[rd_i] = *reinterpret_cast<const u32x4*>(np + rd_i * 1024);
          const int nv = (i + 1) * V / n - i * V / n;
          for (int k = 0; k < nv; ++k, ++fv_i) fv[fv_i & 7] = __builtin_fmaf(fv[fv_i & 7], fm, fv[fv_i & 7]);
          __builtin_amdgcn_sched_barrier(0);
          if (DMA && DPOS == i) { dma_now(); __builtin_amdgcn_sched_barrier(0); }
        };
        if constexpr (SHAPE == 0) {
          // fragments [a hi, a lo, b hi, b lo]; per pair: lo x hi, hi x lo, hi x hi (small terms first)
          f32x16& X = acc[SHAPE == 0 ? m : 0]; f32x16& Y = acc[SHAPE == 0 ? ((m + 4) & 7) : 0];
          X = mma32(FB[1], XB[s][0], X); gap(0, 6); X = mma32(FB[0], XB[s][1], X); gap(1, 6); X = mma32(FB[0], XB[s][0], X); gap(2, 6);
          Y = mma32(FB[3], XB[s][0], Y); gap(3, 6); Y = mma32(FB[2], XB[s][1], Y); gap(4, 6); Y = mma32(FB[2], XB[s][0], Y); gap(5, 6);
        } else {
          // a fragment = 16 features x 32 k; B registers of a k32 step: token half t in XB[2 * (s >> 1) + t] (same 128 registers)
          const int s2 = s & 14;
#pragma unroll
          for (int p = 0; p < 2; ++p) {                      // the two token halves alternate: no MFMA reads the accumulator of the one before it
            f32x4& c0 = a4[4 * m + 2 * p]; f32x4& c1 = a4[4 * m + 2 * p + 1];
            c0 = mma16(FB[2 * p + 1], XB[s2][0], c0); gap(6 * p, 12); c1 = mma16(FB[2 * p + 1], XB[s2 + 1][0], c1); gap(6 * p + 1, 12);
            c0 = mma16(FB[2 * p], XB[s2][1], c0); gap(6 * p + 2, 12); c1 = mma16(FB[2 * p], XB[s2 + 1][1], c1); gap(6 * p + 3, 12);
            c0 = mma16(FB[2 * p], XB[s2][0], c0); gap(6 * p + 4, 12); c1 = mma16(FB[2 * p], XB[s2 + 1][0], c1); gap(6 * p + 5, 12);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) { stamps[((long)blockIdx.x * 4 + wave) * 2] = t1 - t0; stamps[((long)blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < (SHAPE == 0 ? 8 : 1); ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) sum += acc[i][e];
#pragma unroll
  for (int i = 0; i < (SHAPE == 0 ? 1 : 32); ++i) sum += (a4[i][0] + a4[i][1]) + (a4[i][2] + a4[i][3]);
#pragma unroll
  for (int v = 0; v < 8; ++v) sum += fv[v];
  out[(long)blockIdx.x * 256 + tid] = sum;
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }

template <int SHAPE, int V, bool DMA, int DPOS = -1>
static void run(const char* label, const unsigned short* w, const unsigned short* xb, float* out, unsigned long long* stamps, int ncu, double secs) {
  auto k = probe_kernel<SHAPE, V, DMA, DPOS>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, RING * SLAB));
  const int n_slabs = 96 * 12;                               // 12 "tiles" per block and launch
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // >= secs of back-to-back launches to settle the clock, then time 20 launches
  float ms = 0.f; double spent = 0.0;
  while (spent < secs) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(ncu), dim3(256), RING * SLAB, 0, w, xb, out, stamps, n_slabs);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    spent += ms * 1e-3;
  }
  CK(hipGetLastError());
  std::vector<unsigned long long> st((size_t)ncu * 8);
  CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> clk, cyc;
  for (int i = 0; i < ncu * 4; ++i) { clk.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 0.1); cyc.push_back((double)st[2 * i]); }
  std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
  const double us = ms * 1e3 / 20.0;
  // FLOPs: per macro-step and wave 6 x 32768 (= 12 x 16384), 8 macro-steps per slab, 4 waves; fp32-equivalent = / 3
  const double flop = (double)ncu * 4 * n_slabs * 8 * 6 * 32768.0;
  const double cyc_macro = cyc[cyc.size() / 2] / ((double)n_slabs * 8);
  printf("%-46s %8.1f us  %7.1f TF fp16 = %6.1f TF fp32-equiv  clock %.3f GHz (min %.3f max %.3f)  %6.1f cycles / macro-step (ideal 192)\n",
         label, us, flop / us * 1e-6, flop / us * 1e-6 / 3.0, clk[clk.size() / 2], clk.front(), clk.back(), cyc_macro);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 2.5;
  const bool zeros = argc > 2 && !strcmp(argv[2], "zeros");
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("device %s, %d CUs, data %s, %.1f s settle per variant\n", prop.name, ncu, zeros ? "zeros" : "random", secs);
  std::mt19937 rng(7); std::normal_distribution<float> nd(0.f, 1.f);
  // weight planes: hi ~ N(0, 1) x 2^9 (the packed planes sit in [2^10, 2^11) at the maximum), lo = a 2^-11 remainder; B planes likewise at 2^4
  std::vector<unsigned short> hw((size_t)96 * SLAB / 2), hx((size_t)4 * 32 * 64 * 8);
  for (size_t i = 0; i < hw.size(); ++i) { const bool lo = (i / 512) & 1; hw[i] = zeros ? 0 : f2h(nd(rng) * (lo ? 0.25f : 512.f)); }
  for (size_t i = 0; i < hx.size(); ++i) { const bool lo = (i / 512) & 1; hx[i] = zeros ? 0 : f2h(nd(rng) * (lo ? 0.008f : 16.f)); }
  unsigned short *w, *xb; float* out; unsigned long long* stamps;
  CK(hipMalloc(&w, hw.size() * 2)); CK(hipMalloc(&xb, hx.size() * 2)); CK(hipMalloc(&out, (size_t)ncu * 256 * 4)); CK(hipMalloc(&stamps, (size_t)ncu * 8 * 8));
  CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(xb, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
#define RUN(S, V, D, L) run<S, V, D>(L, w, xb, out, stamps, ncu, secs)
  RUN(0, 0, false, "32x32x16  mfma + lds reads");
  RUN(1, 0, false, "16x16x32  mfma + lds reads");
  RUN(0, 0, true,  "32x32x16  + LDS-DMA ring refill");
  RUN(1, 0, true,  "16x16x32  + LDS-DMA ring refill");
  RUN(0, 10, true, "32x32x16  + DMA + 10 VALU / macro-step (bwd)");
  RUN(1, 10, true, "16x16x32  + DMA + 10 VALU / macro-step (bwd)");
  RUN(0, 18, true, "32x32x16  + DMA + 18 VALU / macro-step (fwd)");
  RUN(1, 18, true, "16x16x32  + DMA + 18 VALU / macro-step (fwd)");
  if (argc > 3) {                                            // DMA placement sweep on the 16-wide shape (the piece behind MFMA p of the macro-step's 12)
    run<1, 10, true, 3>("16x16x32  + DMA behind MFMA 3  + 10 VALU", w, xb, out, stamps, ncu, secs);
    run<1, 10, true, 7>("16x16x32  + DMA behind MFMA 7  + 10 VALU", w, xb, out, stamps, ncu, secs);
    run<1, 10, true, 9>("16x16x32  + DMA behind MFMA 9  + 10 VALU", w, xb, out, stamps, ncu, secs);
    run<1, 10, true, 11>("16x16x32  + DMA behind MFMA 11 + 10 VALU", w, xb, out, stamps, ncu, secs);
    run<1, 18, true, 9>("16x16x32  + DMA behind MFMA 9  + 18 VALU", w, xb, out, stamps, ncu, secs);
    run<1, 18, true, 11>("16x16x32  + DMA behind MFMA 11 + 18 VALU", w, xb, out, stamps, ncu, secs);
  }
  RUN(0, 0, false, "32x32x16  mfma + lds reads (repeat)");
  RUN(1, 0, false, "16x16x32  mfma + lds reads (repeat)");
  return 0;
}

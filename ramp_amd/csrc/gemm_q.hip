// fp16x3 GEMM with both operands through an LDS ring filled by LDS-DMA (global_load_lds): no vector ALU work, no
// staging registers and no duplicated fragment fetches in the main loop.
//
// Why (measured with the ablation twin of the register-staged kernel, ramp_amd/tools/gemm_ablate.py, 393216-row shapes):
// that kernel's loop runs at 330-350 TFLOP/s on long-K shapes while the same loop WITHOUT its operand traffic runs at
// 520-550; switching off the activation staging (global load -> scale -> split into two fp16 planes -> ds_write, ~60
// vector instructions per 24 MFMAs) buys 14-37 %, switching off the per-wave weight-fragment loads (every wave pulls its
// own fragments through the CU's single 64 B/clk vector-memory path: ~60 B/clk demanded at full MFMA rate) buys 13-26 %.
// Here the activations arrive ALREADY split (two scaled fp16 planes, row-major [2][M][K], written by the producing
// kernel or by launch_split_planes), and a 128 x 256 tile shared by eight waves takes every A row-slab once per 256
// columns and every weight fragment once per 128 rows: 24 KB per 16-deep K step and block = ~31 B/clk.
//
//   ring of QR stages; stage = A planes [2][128 rows][32 B] (16-byte slots XOR-swizzled by (row >> 3) & 1: conflict-free
//   32x32x16 fragment reads) + weight fragments [8 column groups][2 planes][1 KB] copied verbatim from the
//   fragment-packed weights (launch_pack_h3), so a wave reads its B operand at lane * 16.
//   iteration g:  s_waitcnt vmcnt(own LDS-DMA of stage g+1 landed) lgkmcnt(0) ; s_barrier ; LDS-DMA of stage g+QR into
//                 the slot stage g just vacated ; ds_read fragments of stage g+1 ; 12 MFMAs of stage g
//   The stage stream runs on across tile boundaries (the loader walks the same tile list ahead of the MFMAs), so a
//   tile's first stages are in LDS before its predecessor's epilogue starts, and the epilogue's stores sit BEHIND them in
//   the (in-order) vmcnt queue: the waits of a tile's first QR-1 stages allow exactly those stores to stay in flight.
#include "common.h"

#include <algorithm>

namespace ramp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

constexpr int QBM = 128, QBN = 256, QR = 5;
constexpr int QST_A = 2 * QBM * 32;                 // bytes of the A planes of one stage (8 KB)
constexpr int QST = QST_A + 8 * 2 * 1024;           // + weight fragments (16 KB)
constexpr int QSLD = 68;                            // floats per epilogue scratch row
constexpr int QSCR = 16 * QSLD * 4;                 // wave-private epilogue scratch: 16 rows x 64 columns
constexpr size_t Q_LDS = (size_t)QR * QST + 8 * QSCR;
static_assert(Q_LDS <= 160 * 1024, "LDS budget");
constexpr int Q_EPI_STORES = 16;                    // global stores per wave and tile (64 x 64 fp32 = 16 x 1 KB)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* glb_ptr_t;

__device__ __forceinline__ void glds16(const void* src, char* dst) {
  __builtin_amdgcn_global_load_lds((glb_ptr_t)(uintptr_t)src, (lds_ptr_t)(unsigned)(uintptr_t)dst, 16, 0, 0);
}

// EPI_LINEAR only: C = acc * oscale + bias + rowbias + resid.  M must be a multiple of 128 and N of 256 (the caller
// falls back to the register-staged kernel otherwise), so every store is unconditional and the per-tile store count the
// vmcnt bookkeeping relies on is a constant.
// STAG: the two waves of every SIMD (waves w and w + 4 of the block) alternate -- one reads its next fragments and issues
// LDS-DMA while the other owns the matrix pipe -- by running waves 4..7 one barrier phase behind waves 0..3 (two phases
// per stage).  A wave then confirms its own LDS-DMA one stage earlier (stage g + 2 at phase L(g)), because the other
// half reads a stage half a phase before / after it does.
template <bool STAG>
__global__ __launch_bounds__(512)
void gemm_h3q_kernel(GemmArgs a, int tiles_n, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int bid = blockIdx.x, nb = gridDim.x;
  const int xcd = bid & 7, slot = bid >> 3, bpx = nb >> 3;
  const long xlo = (long)xcd * n_tiles / 8, xhi = (long)(xcd + 1) * n_tiles / 8;
  const int t_begin = (int)xlo + slot, t_end = (int)xhi, t_step = bpx;
  if (t_begin >= t_end) return;
  const int nk = a.K / 16;

  // the scale the producer applied to the planes (same rule as the register-staged kernel's loader)
  float s_a = 1.f;
  {
    const float mx = a.a_absmax_in ? *a.a_absmax_in : 0.f;
    if (mx > 0.f) {
      int eb = (int)((__builtin_bit_cast(unsigned, mx) >> 23) & 0xffu);
      int sb = 259 - eb;
      sb = sb < 1 ? 1 : (sb > 254 ? 254 : sb);
      s_a = __builtin_bit_cast(float, (unsigned)sb << 23);
    }
  }
  const float oscale = a.w_scale_inv / s_a;

  // ---- loader: one A piece (32 rows of one plane) and two weight-fragment blocks per wave and stage -------------
  int ld_tile = t_begin, ld_ks = 0, ld_slot = 0;
  const int a_plane = wave >> 2, a_rb = wave & 3;
  const int a_row = a_rb * 32 + (lane >> 1);
  const int a_k8 = ((lane & 1) ^ ((a_row >> 3) & 1)) * 8;                   // source slot of this lane's 16 bytes
  const unsigned short* a_src = nullptr; const char* b_src = nullptr;
  auto setup_tile = [&](int tile) {
    const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
    const long m = (long)tile_m * QBM + a_row;
    a_src = a.Ap + (long)a_plane * a.ap_plane + m * a.K + a_k8;
    b_src = reinterpret_cast<const char*>(a.Wx) + ((long)(tile_n * 8 + wave) * nk) * 2048 + lane * 16;
  };
  auto issue_stage = [&]() {
    char* st = smem + ld_slot * QST;
    glds16(a_src + ld_ks * 16, st + a_plane * 4096 + a_rb * 1024);
    glds16(b_src + (long)ld_ks * 2048, st + QST_A + wave * 2048);
    glds16(b_src + (long)ld_ks * 2048 + 1024, st + QST_A + wave * 2048 + 1024);
    if (++ld_ks == nk) { ld_ks = 0; if (ld_tile + t_step < t_end) { ld_tile += t_step; setup_tile(ld_tile); } }
    ld_slot = ld_slot + 1 == QR ? 0 : ld_slot + 1;
  };

  // ---- fragment reads ---------------------------------------------------------------------------------------------
  const int r = lane & 31, h = lane >> 5;
  int a_off[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int row = wm * 64 + mi * 32 + r;
    a_off[mi] = row * 32 + ((h ^ ((row >> 3) & 1)) << 4);
  }
  const int b_off = QST_A + wn * 4096 + lane * 16;                            // + ni * 2048 + plane * 1024
  auto read_frags = [&](u32x4 (&fa)[2][2], u32x4 (&fb)[2][2], int slot_r) {
    const char* st = smem + slot_r * QST;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[p][i] = *reinterpret_cast<const u32x4*>(st + p * 4096 + a_off[i]);
        fb[p][i] = *reinterpret_cast<const u32x4*>(st + b_off + i * 2048 + p * 1024);
      }
  };

  float* Sw = reinterpret_cast<float*>(smem + QR * QST + wave * QSCR);

  // ---- prologue: QR stages in flight, fragments of stage 0 in registers -------------------------------------------
  setup_tile(ld_tile);
#pragma unroll 1
  for (int i = 0; i < QR; ++i) issue_stage();
  if (STAG) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (QR - 2)) : "memory");      // own shares of stages 0 and 1
  else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (QR - 1)) : "memory");
  __builtin_amdgcn_s_barrier();
  u32x4 fa[2][2][2], fb[2][2][2];                     // [buffer][plane][mi / ni]
  read_frags(fa[0], fb[0], 0);
  int rd_slot = 1;                                     // slot of the stage whose fragments are read next
  bool after_epi = false;
  if (STAG && wm == 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }   // half a stage behind

  for (int tile = t_begin; tile < t_end; tile += t_step) {
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

#pragma unroll 1
    for (int ks = 0; ks < nk; ks += 2) {
      // two stages per trip so that the fragment double buffer is indexed statically
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        // stage g+1 (STAG: g+2) landed (this wave's share; the barrier makes it everyone's), reads of stage g retired
        constexpr int LEAD = STAG ? QR - 3 : QR - 2;
        if (after_epi && ks + u <= LEAD) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(3 * LEAD + Q_EPI_STORES) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(3 * LEAD) : "memory");
        __builtin_amdgcn_s_barrier();
        issue_stage();                                 // stage g + QR -> the slot stage g has just left
        read_frags(fa[u ^ 1], fb[u ^ 1], rd_slot);
        rd_slot = rd_slot + 1 == QR ? 0 : rd_slot + 1;
        if (STAG) {                                    // second phase of the stage: this half computes, the other loads
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_s_setprio(1);
        }
        // small terms first: h2 h1', h1 h2', h1 h1'
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fa[u][PA[t3]][mi]),
                                                                   __builtin_bit_cast(half8, fb[u][PB[t3]][ni]), acc[mi][ni], 0, 0, 0);
        if (STAG) __builtin_amdgcn_s_setprio(0);
      }
    }

    // ---- epilogue: four passes of 16 rows x 64 columns through the wave's own scratch -------------------------------
    const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
    const int rl0 = lane >> 4, c = (lane & 15) * 4;          // 4 row groups x 16 lanes per row
    const int n = tile_n * QBN + wn * 64 + c;
    f32x4 b4 = {0, 0, 0, 0};
    if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + n);
    // every residual row is requested BEFORE the first store: vmcnt is in order, a load behind a store waits for its ack
    f32x4 pre[4][4];
    if (a.resid) {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
        for (int p = 0; p < 4; ++p)
          pre[q4][p] = *reinterpret_cast<const f32x4*>(a.resid + (long)(tile_m * QBM + wm * 64 + q4 * 16 + p * 4 + rl0) * a.ldr + n);
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int m0 = tile_m * QBM + wm * 64 + mi * 32 + hf * 16;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int reg = hf * 8 + q;                      // rows (reg & 3) + 8 (reg >> 2) + 4 h of the 32: 0..15 for reg < 8
            Sw[((q & 3) + 8 * (q >> 2) + 4 * h) * QSLD + ni * 32 + r] = acc[mi][ni][reg];
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int rl = p * 4 + rl0, m = m0 + rl;
          f32x4 v = *reinterpret_cast<const f32x4*>(Sw + rl * QSLD + c);
          v = v * oscale + b4;
          if (a.rowbias) v += *reinterpret_cast<const f32x4*>(a.rowbias + (long)a.rowvar[a.row0 + m / a.L] * a.rb_stride + n);
          if (a.resid) v += pre[mi * 2 + hf][p];
          *reinterpret_cast<f32x4*>(a.C + (long)m * a.ldc + n) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    after_epi = true;
  }
  if (STAG && wm == 0) __builtin_amdgcn_s_barrier();     // the phase waves 4..7 still owe
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the loader ran ahead: no LDS-DMA may outlive the block
}

// fp32 [M][lda] -> two scaled fp16 planes, row-major [2][M][K] (plane stride M * K), the operand format of
// gemm_h3q_kernel; same delayed scale, maximum recording and range guard as the register-staged kernel's loader.
__global__ __launch_bounds__(256)
void split_planes_kernel(const float* __restrict__ A, int lda, unsigned short* __restrict__ out, long plane, int M, int K,
                         const float* __restrict__ absmax_in, float* __restrict__ absmax_out, int* __restrict__ range_flag,
                         int site_id) {
  float s_a = 1.f;
  {
    const float mx = absmax_in ? *absmax_in : 0.f;
    if (mx > 0.f) {
      int eb = (int)((__builtin_bit_cast(unsigned, mx) >> 23) & 0xffu);
      int sb = 259 - eb;
      sb = sb < 1 ? 1 : (sb > 254 ? 254 : sb);
      s_a = __builtin_bit_cast(float, (unsigned)sb << 23);
    }
  }
  const int k4n = K / 4;
  const long total = (long)M * k4n;
  float amax = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / k4n; const int k4 = (int)(i - m * k4n);
    f32x4 v = *reinterpret_cast<const f32x4*>(A + m * lda + k4 * 4);
    amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fmaxf(fabsf(v[2]), fabsf(v[3])), amax));
    v *= s_a;
    const half2v h0 = __builtin_convertvector(f32x2{v[0], v[1]}, half2v), h1 = __builtin_convertvector(f32x2{v[2], v[3]}, half2v);
    const f32x2 r0 = __builtin_convertvector(h0, f32x2), r1 = __builtin_convertvector(h1, f32x2);
    const half2v l0 = __builtin_convertvector(f32x2{v[0] - r0[0], v[1] - r0[1]}, half2v);
    const half2v l1 = __builtin_convertvector(f32x2{v[2] - r1[0], v[3] - r1[1]}, half2v);
    unsigned short* o = out + m * K + k4 * 4;
    *reinterpret_cast<u32x2*>(o) = u32x2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
    *reinterpret_cast<u32x2*>(o + plane) = u32x2{__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1)};
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
  if ((threadIdx.x & 63) == 0) {
    record_amax(absmax_out, amax);
    if (range_flag && (!(amax * s_a < 60000.f) || (amax > 0.f && amax * s_a < 0.125f))) atomicMax(range_flag, site_id + 1);
  }
}

int launch_split_planes(const float* A, int lda, unsigned short* out, long plane, int M, int K, const float* absmax_in,
                        float* absmax_out, int* range_flag, int site_id, hipStream_t s) {
  RAMP_REQUIRE(A && out && M > 0 && K > 0 && K % 4 == 0 && lda % 4 == 0, "bad split dims");
  const long total = (long)M * (K / 4);
  hipLaunchKernelGGL(split_planes_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, s, A, lda, out,
                     plane, M, K, absmax_in, absmax_out, range_flag, site_id);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

bool gemm_h3q_applicable(const GemmArgs& a) {
  return a.Ap != nullptr && a.wx_packed == 2 && a.epi == EPI_LINEAR && a.taps == 1 && a.M % QBM == 0 && a.N % QBN == 0 &&
         a.K % 32 == 0 && a.K / 16 >= QR && !a.A2 && !a.C2 && !a.resid2 && !a.Amul && a.a_stride == 1 && a.c_rstride == 1 &&
         a.c_roff == 0 && a.shift0 == 0;
}

int launch_gemm_h3q(const GemmArgs& a, hipStream_t s) {
  RAMP_REQUIRE(gemm_h3q_applicable(a), "shape not covered by the LDS-DMA GEMM");
  RAMP_REQUIRE(a.ldc % 4 == 0 && (a.resid == nullptr || a.ldr % 4 == 0) && a.ap_plane >= (long)a.M * a.K, "bad leading dimensions");
  const int tiles_m = a.M / QBM, tiles_n = a.N / QBN, n_tiles = tiles_m * tiles_n;
  const int slots = 256;                                   // one 8-wave block per CU
  const int rounds = (n_tiles + slots - 1) / slots;
  const int nb = std::min((((n_tiles + rounds - 1) / rounds + 7) / 8) * 8, slots);
  if (a.tile_pref == 1) hipLaunchKernelGGL(gemm_h3q_kernel<false>, dim3(nb), dim3(512), Q_LDS, s, a, tiles_n, n_tiles);
  else hipLaunchKernelGGL(gemm_h3q_kernel<true>, dim3(nb), dim3(512), Q_LDS, s, a, tiles_n, n_tiles);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

int init_gemm_q_attributes() {
  RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h3q_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)Q_LDS));
  RAMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h3q_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)Q_LDS));
  return 0;
}

}  // namespace ramp

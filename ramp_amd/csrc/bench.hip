// ramp_bench_gemm / ramp_stress_gemm: the micro-benchmark of every kernel family on synthetic operands and the bitwise stress hook behind
// the soak tests.  DIAGNOSTICS: this translation unit is linked into ramp_amd/lib/libramp_hip_tools.so only (the product library
// libramp_hip.so exports no ramp_bench_* / ramp_stress_*); ramp_amd._lib.load_tools() binds it for tests/ and ramp_amd/tools/.
#include "engine_util.h"
#include "../../include/ramp_hip_tools.h"

extern "C" {

// micro-benchmark of one GEMM shape on a named kernel: packs once, `warmup` + `iters` back-to-back launches on `stream`,
// HIP events around the timed ones.  flags: 1 bias, 2 residual, 4 GEGLU-forward epilogue (N = 2F, writes the F-wide
// product too), 8 A-multiplier operand (K = 2 * period).  Operands are allocated and filled here (uniform [-1, 1)).
__global__ void fill_uniform_kernel(float* p, long n, unsigned seed, float scale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = ((float)(h >> 8) * (1.f / 8388608.f) - 1.f) * scale;
  }
}
// ---- stress hook of the micro-benchmark: every launch's output against the first one's, bit for bit ------------------------
namespace {
__global__ void stress_cmp_kernel(const unsigned* __restrict__ a, const unsigned* __restrict__ b, long n, unsigned long long* mism) {
  unsigned long long c = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) c += a[i] != b[i];
  for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(mism, c);
}
__global__ void stress_err_kernel(const float* __restrict__ a, const float* __restrict__ ref, long n, unsigned* out /* [max |a - ref|, max |ref|] as float bits */) {
  float e = 0.f, r = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { e = fmaxf(e, fabsf(a[i] - ref[i])); r = fmaxf(r, fabsf(ref[i])); }
  for (int m = 32; m >= 1; m >>= 1) { e = fmaxf(e, __shfl_xor(e, m)); r = fmaxf(r, __shfl_xor(r, m)); }
  if ((threadIdx.x & 63) == 0) { atomicMax(out, __builtin_bit_cast(unsigned, e)); atomicMax(out + 1, __builtin_bit_cast(unsigned, r)); }
}
struct StressHook {
  bool capture_ref = false;      // this run only provides the reference output (the exact-fp32 kernel)
  float* first = nullptr; float* ref = nullptr; long n = 0; long launches = 0;
  unsigned long long* mism = nullptr; unsigned* err = nullptr;
  // after a launch that wrote `out` (n floats)
  int check(const float* out, long nf, hipStream_t s) {
    if (capture_ref) {
      if (!ref) { RAMP_HIP_CHECK(hipMalloc(&ref, nf * 4)); n = nf; }
      RAMP_HIP_CHECK(hipMemcpyAsync(ref, out, nf * 4, hipMemcpyDeviceToDevice, s));
      return 0;
    }
    if (!mism) { RAMP_HIP_CHECK(hipMalloc(&mism, 16)); RAMP_HIP_CHECK(hipMemsetAsync(mism, 0, 16, s)); err = reinterpret_cast<unsigned*>(mism) + 2; }
    if (!first) {
      RAMP_HIP_CHECK(hipMalloc(&first, nf * 4));
      RAMP_HIP_CHECK(hipMemcpyAsync(first, out, nf * 4, hipMemcpyDeviceToDevice, s));
      if (ref && n == nf) hipLaunchKernelGGL(stress_err_kernel, dim3(2048), dim3(256), 0, s, out, ref, nf, err);
    } else {
      hipLaunchKernelGGL(stress_cmp_kernel, dim3(2048), dim3(256), 0, s, reinterpret_cast<const unsigned*>(out),
                         reinterpret_cast<const unsigned*>(first), nf, mism);
    }
    ++launches;
    RAMP_HIP_CHECK(hipGetLastError());
    return 0;
  }
  ~StressHook() { if (first) (void)hipFree(first); if (ref) (void)hipFree(ref); if (mism) (void)hipFree(mism); }
};
thread_local StressHook* g_stress = nullptr;
#define STRESS(ptr_, n_, s_) do { if (g_stress) CK(g_stress->check((ptr_), (long)(n_), (s_))); } while (0)
}  // namespace

int ramp_bench_gemm(int32_t M, int32_t N, int32_t K, int32_t taps, int32_t L, int32_t mode, int32_t flags,
                    int32_t warmup, int32_t iters, float* avg_us, void* stream) {
  RAMP_REQUIRE(avg_us && iters > 0 && M > 0 && N > 0 && K > 0 && taps >= 1 && mode >= 0 && mode <= 17, "bad arguments");
  if (mode == 17) {                                    // tkw.hip: wide k = 5 convolution on sample-owning blocks; flags 1: GroupNorm + Mish epilogue (forward),
                                                       // 2: GroupNorm-backward operand (input gradient), 4: dir = -1, 8: residual, 16: bias
    hipStream_t sw = as_stream(stream);
    DevArena arw;
    const bool epi = flags & 1, pro = flags & 2;
    RAMP_REQUIRE(taps == 5 && tkw_applicable(M, L, N, K, pro, epi), "mode 17: k = 5, C_out in {128, 256, 512}, L >= 3 dividing 96");
    float* X = arw.alloc((size_t)M * K); float* Y = arw.alloc((size_t)M * N); float* R = arw.alloc((size_t)M * N); float* Cs = arw.alloc((size_t)M * std::max(N, K));
    float* W = arw.alloc((size_t)5 * N * K); float* b = arw.alloc(N); float* gm = arw.alloc(std::max(N, K)); float* bt = arw.alloc(std::max(N, K));
    float* st = arw.alloc((size_t)(M / L) * 16); float* sl = arw.alloc(4); float* tmp = arw.alloc((size_t)M * K);
    unsigned short* pl = reinterpret_cast<unsigned short*>(arw.alloc((size_t)5 * N * K + 4));
    RAMP_REQUIRE(X && Y && R && Cs && W && b && gm && bt && st && sl && tmp && pl, "hipMalloc failed");
    auto fill = [&](float* p, size_t n, unsigned seed, float scv) { hipLaunchKernelGGL(fill_uniform_kernel, dim3(2048), dim3(256), 0, sw, p, (long)n, seed, scv); };
    fill(X, (size_t)M * K, 1u, 1.f); fill(R, (size_t)M * N, 4u, 1.f); fill(W, (size_t)5 * N * K, 2u, 1.f / std::sqrt(5.f * K)); fill(b, N, 5u, 1.f);
    fill(gm, std::max(N, K), 6u, 1.f); fill(bt, std::max(N, K), 7u, 0.5f); fill(Cs, (size_t)M * std::max(N, K), 8u, 1.f);
    CK(init_tkw_attributes());
    float scp = 1.f; { int e; std::frexp(std::ldexp(1.f, 10) * std::sqrt(5.f * K), &e); scp = std::ldexp(1.f, e - 1); }
    CK(launch_pack_h3(W, pl, (long)5 * N, K, scp, sw));
    if (pro) {      // statistics of the tensor the backward normalises again (any consistent mean / rstd do)
      GnArgs g; g.x = Cs; g.gamma = gm; g.beta = bt; g.y = tmp; g.stats = st; g.R = M / L; g.L = L; g.C = K; g.eps = 1e-5f; g.mish = 1;
      CK(launch_gn_fwd(g, sw));
    }
    const float one[4] = {pro ? 8.f : 1.f, 0.f, 0.f, 0.f};
    RAMP_HIP_CHECK(hipMemcpyAsync(sl, one, sizeof(one), hipMemcpyHostToDevice, sw));
    TkwArgs t; t.M = M; t.L = L; t.N = N; t.K = K; t.dir = (flags & 4) ? -1 : 1; t.X = X; t.ldx = K; t.K1 = K; t.W = pl; t.wsi = 1.f / scp; t.Y = Y; t.ldy = N; t.N1 = N;
    if ((flags & 16) || epi) t.bias = b;
    if (flags & 8) { t.resid = R; t.ldr = N; }
    if (pro) { t.gn_c = Cs; t.gn_stats = st; t.gn_gamma = gm; t.gn_beta = bt; }
    if (epi) { t.Cst = Cs; t.stats = st; t.gamma = gm; t.beta = bt; t.tbias = b; }
    t.amax_in = sl; t.amax_out = sl + 1; t.range_flag = reinterpret_cast<int*>(sl + 2); t.ablate = (flags >> 8) & 7;
    for (int i = 0; i < warmup; ++i) CK(launch_tkw(t, sw));
    hipEvent_t e0, e1;
    RAMP_HIP_CHECK(hipEventCreate(&e0)); RAMP_HIP_CHECK(hipEventCreate(&e1));
    RAMP_HIP_CHECK(hipEventRecord(e0, sw));
    int rcw = 0;
    for (int i = 0; i < iters && rcw == 0; ++i) { rcw = launch_tkw(t, sw); if (rcw == 0) STRESS(Y, (size_t)M * N, sw); }
    RAMP_HIP_CHECK(hipEventRecord(e1, sw));
    RAMP_HIP_CHECK(hipEventSynchronize(e1));
    float msw = 0.f;
    RAMP_HIP_CHECK(hipEventElapsedTime(&msw, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_us = msw * 1e3f / iters;
    return rcw;
  }
  if (mode == 15 || mode == 16) {                      // attention backward + d(ln1) + LN1 backward: abl_kernel (15, atl.hip) / atb_kernel + tklb_kernel (16)
    hipStream_t sb = as_stream(stream);
    DevArena arb;
    RAMP_REQUIRE(L >= 1 && M % L == 0, "mode 15 / 16: M must be whole samples of L tokens");
    float* Q = arb.alloc((size_t)M * 768); float* D = arb.alloc((size_t)M * 256); float* G = arb.alloc((size_t)M * 768);
    float* Z = arb.alloc((size_t)M * 256); float* Ad = arb.alloc((size_t)M * 256); float* Y = arb.alloc((size_t)M * 256);
    float* Wq = arb.alloc((size_t)256 * 768); float* gam = arb.alloc(256); float* slots = arb.alloc(4);
    unsigned short* ws = reinterpret_cast<unsigned short*>(arb.alloc((size_t)256 * 768 + 4));
    unsigned short* pl = reinterpret_cast<unsigned short*>(arb.alloc((size_t)256 * 768 + 4));
    RAMP_REQUIRE(Q && D && G && Z && Ad && Y && Wq && gam && slots && ws && pl, "hipMalloc failed");
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(2048), dim3(256), 0, sb, Q, (long)M * 768, 1u, 1.5f);
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(2048), dim3(256), 0, sb, D, (long)M * 256, 3u, 1.f);
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(2048), dim3(256), 0, sb, Z, (long)M * 256, 5u, 1.f);
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(2048), dim3(256), 0, sb, Ad, (long)M * 256, 7u, 1.f);
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(64), dim3(256), 0, sb, Wq, 256l * 768, 9u, 1.f / 16.f);
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(1), dim3(256), 0, sb, gam, 256l, 11u, 1.f);
    CK(init_atl_attributes()); CK(init_tkl_attributes());
    CK(abl_pack(Wq, 16384.f, ws, sb));
    CK(launch_pack_h3(Wq, pl, 256, 768, 16384.f, sb));
    const float host[4] = {4.f, 0.f, 0.f, 0.f};
    RAMP_HIP_CHECK(hipMemcpyAsync(slots, host, sizeof(host), hipMemcpyHostToDevice, sb));
    AblArgs t; t.M = M; t.L = L; t.QKV = Q; t.dO = D; t.W = ws; t.Z = Z; t.add = Ad; t.ln_g = gam; t.Y = Y;
    t.no_park = (flags >> 16) & 1;                       // flags bit 16: the round-4 kernel that fetches k a second time (A/B twin, same bits)
    t.amax_in = slots; t.amax_out = slots + 1; t.wsi = 1.f / 16384.f; t.range_flag = nullptr;
    AtbArgs tb; tb.M = M; tb.L = L; tb.QKV = Q; tb.dO = D; tb.dQKV = G;
    TklbArgs tl; tl.M = M; tl.X = G; tl.Z = Z; tl.add = Ad; tl.Y = Y; tl.W = pl; tl.ln_g = gam;
    tl.amax_in = slots; tl.amax_out = slots + 1; tl.wsi = 1.f / 16384.f;
    unsigned long long* stamps = nullptr;
    if (flags & 256) {
      RAMP_REQUIRE(mode == 15, "stamps: mode 15");
      stamps = reinterpret_cast<unsigned long long*>(arb.alloc(256 * 4 * 10 * 2));
      RAMP_REQUIRE(stamps, "hipMalloc failed");
      RAMP_HIP_CHECK(hipMemsetAsync(stamps, 0, 256 * 4 * 10 * 8, sb));
      t.stamps = stamps;
    }
    auto go = [&]() -> int { if (mode == 15) return launch_abl(t, sb); int rc = launch_atb(tb, sb); return rc ? rc : launch_tklb(tl, sb); };
    for (int i = 0; i < warmup; ++i) CK(go());
    hipEvent_t e0, e1;
    RAMP_HIP_CHECK(hipEventCreate(&e0)); RAMP_HIP_CHECK(hipEventCreate(&e1));
    RAMP_HIP_CHECK(hipEventRecord(e0, sb));
    int rcb = 0;
    for (int i = 0; i < iters && rcb == 0; ++i) { rcb = go(); if (rcb == 0) STRESS(Y, (size_t)M * 256, sb); }
    RAMP_HIP_CHECK(hipEventRecord(e1, sb));
    RAMP_HIP_CHECK(hipEventSynchronize(e1));
    float msb = 0.f;
    RAMP_HIP_CHECK(hipEventElapsedTime(&msb, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_us = msb * 1e3f / iters;
    if (rcb == 0 && stamps) {
      std::vector<unsigned long long> hst(256 * 4 * 10);
      RAMP_HIP_CHECK(hipMemcpy(hst.data(), stamps, hst.size() * 8, hipMemcpyDeviceToHost));
      double sm[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; int nw = 0;
      for (int w = 0; w < 1024; ++w) if (hst[w * 10 + 8]) { ++nw; for (int j = 0; j < 10; ++j) sm[j] += (double)hst[w * 10 + j]; }
      const int n_tiles = (M + 191) / 192;
      const double tiles = std::max(1, nw) * (double)((n_tiles + 255) / 256);
      fprintf(stderr, "[abl stamps] per wave tile (s_memtime ticks, %d waves): head-start waits %.0f, S^T + softmax %.0f, d(o) / v waits %.0f, dP + dS + planes + P turned %.0f, "
              "turn + contract (x 24) %.0f, slab waits + barriers (x 48) %.0f, slab bodies %.0f, epilogue + loop tops %.0f; whole kernel %.0f per tile; shader clock %.0f MHz\n",
              nw, sm[0] / tiles, sm[1] / tiles, sm[2] / tiles, sm[3] / tiles, sm[4] / tiles, sm[5] / tiles, sm[6] / tiles, sm[7] / tiles, sm[8] / tiles,
              sm[9] > 0 ? sm[8] / sm[9] * 100.0 : 0.0);
    }
    return rcb;
  }
  if (mode == 13 || mode == 14) {                      // attention backward: atb_kernel (13, atk.hip) / attn2_bwd_kernel (14, attention.hip); L = tokens per sample
    hipStream_t sb = as_stream(stream);
    DevArena arb;
    RAMP_REQUIRE(L >= 1 && M % L == 0, "mode 13 / 14: M must be whole samples of L tokens");
    float* Q = arb.alloc((size_t)M * 768); float* D = arb.alloc((size_t)M * 256); float* G = arb.alloc((size_t)M * 768);
    RAMP_REQUIRE(Q && D && G, "hipMalloc failed");
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(2048), dim3(256), 0, sb, Q, (long)M * 768, 1u, 1.5f);
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(2048), dim3(256), 0, sb, D, (long)M * 256, 3u, 1.f);
    AtbArgs t; t.M = M; t.L = L; t.QKV = Q; t.dO = D; t.dQKV = G;
    auto go = [&]() -> int { return mode == 13 ? launch_atb(t, sb) : launch_attn_bwd(Q, D, G, M / L, L, sb); };
    for (int i = 0; i < warmup; ++i) CK(go());
    hipEvent_t e0, e1;
    RAMP_HIP_CHECK(hipEventCreate(&e0)); RAMP_HIP_CHECK(hipEventCreate(&e1));
    RAMP_HIP_CHECK(hipEventRecord(e0, sb));
    int rcb = 0;
    for (int i = 0; i < iters && rcb == 0; ++i) { rcb = go(); if (rcb == 0) STRESS(G, (size_t)M * 768, sb); }
    RAMP_HIP_CHECK(hipEventRecord(e1, sb));
    RAMP_HIP_CHECK(hipEventSynchronize(e1));
    float msb = 0.f;
    RAMP_HIP_CHECK(hipEventElapsedTime(&msb, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_us = msb * 1e3f / iters;
    return rcb;
  }
  if (mode == 12) {                                    // tkc.hip: k = 5 convolution with C_in, C_out in {32, 64} on sample-owning waves; flags 1 bias, 2 residual, 4 input gradient,
                                                       // 8 GroupNorm + Mish epilogue, 16 GroupNorm-backward operand
    hipStream_t sc = as_stream(stream);
    DevArena arc;
    RAMP_REQUIRE(taps == 5 && tkc_applicable(M, L, N, K, nullptr), "mode 12: k = 5, C in {32, 64}, L >= 8 dividing 48 or 32");
    float* X = arc.alloc((size_t)M * K); float* Y = arc.alloc((size_t)M * N); float* R = arc.alloc((size_t)M * N); float* W = arc.alloc((size_t)5 * N * K);
    float* b = arc.alloc(N); float* sl = arc.alloc(4);
    float* Cs = arc.alloc((size_t)M * 64); float* gm = arc.alloc(64); float* bt = arc.alloc(64); float* st = arc.alloc((size_t)(M / L) * 16);
    unsigned short* pl = reinterpret_cast<unsigned short*>(arc.alloc(tkc_packed_halves(N, K) / 2 + 4));
    RAMP_REQUIRE(X && Y && R && W && b && sl && pl && Cs && gm && bt && st, "hipMalloc failed");
    auto fill = [&](float* p, size_t n, unsigned seed, float scv) { hipLaunchKernelGGL(fill_uniform_kernel, dim3(2048), dim3(256), 0, sc, p, (long)n, seed, scv); };
    fill(X, (size_t)M * K, 1u, 1.f); fill(R, (size_t)M * N, 4u, 1.f); fill(W, (size_t)5 * N * K, 2u, 1.f / 16.f); fill(b, N, 5u, 1.f);
    CK(init_tkc_attributes());
    CK(tkc_pack(W, N, K, 16384.f, pl, sc));
    const float one[4] = {(flags & 16) ? 8.f : 1.f, 0.f, 0.f, 0.f};
    RAMP_HIP_CHECK(hipMemcpyAsync(sl, one, sizeof(one), hipMemcpyHostToDevice, sc));
    TkcArgs t; t.M = M; t.L = L; t.N = N; t.K = K; t.dir = (flags & 4) ? -1 : 1; t.X = X; t.ldx = K; t.W = pl; t.Y = Y; t.ldy = N;
    if (flags & 1) t.bias = b;
    if (flags & 2) { t.resid = R; t.ldr = N; }
    if (flags & 24) { fill(gm, 64, 6u, 1.f); fill(bt, 64, 7u, 0.5f); fill(Cs, (size_t)M * 64, 8u, 1.f); }
    if (flags & 16) {
      GnArgs g; g.x = Cs; g.gamma = gm; g.beta = bt; g.y = Y; g.stats = st; g.R = M / L; g.L = L; g.C = K; g.eps = 1e-5f; g.mish = 1;
      RAMP_REQUIRE(N >= K, "mode 12 with flag 16: the scratch of the statistics pass is the output");
      CK(launch_gn_fwd(g, sc));
      t.gn_c = Cs; t.gn_stats = st; t.gn_gamma = gm; t.gn_beta = bt;
    }
    if (flags & 8) { t.bias = b; t.Cst = Cs; t.stats = st; t.gamma = gm; t.beta = bt; t.tbias = b; }
    t.amax_in = sl; t.amax_out = sl + 1; t.wsi = 1.f / 16384.f; t.range_flag = reinterpret_cast<int*>(sl + 2);
    for (int i = 0; i < warmup; ++i) CK(launch_tkc(t, sc));
    hipEvent_t e0, e1;
    RAMP_HIP_CHECK(hipEventCreate(&e0)); RAMP_HIP_CHECK(hipEventCreate(&e1));
    RAMP_HIP_CHECK(hipEventRecord(e0, sc));
    int rcc = 0;
    for (int i = 0; i < iters && rcc == 0; ++i) { rcc = launch_tkc(t, sc); if (rcc == 0) STRESS(Y, (size_t)M * N, sc); }
    RAMP_HIP_CHECK(hipEventRecord(e1, sc));
    RAMP_HIP_CHECK(hipEventSynchronize(e1));
    float msc = 0.f;
    RAMP_HIP_CHECK(hipEventElapsedTime(&msc, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_us = msc * 1e3f / iters;
    return rcc;
  }
  if (mode == 10 || mode == 11) {                      // atk.hip: self-attention + out-projection in one launch (10) / the pair it replaces:
                                                       // attn2_fwd + token-owning out-projection (11).  L = tokens per sample; flags 1: row-variant constant
    hipStream_t sa = as_stream(stream);
    DevArena ara;
    RAMP_REQUIRE(L >= 1 && M % L == 0, "mode 10 / 11: M must be whole samples of L tokens");
    float* Q = ara.alloc((size_t)M * 768); float* R = ara.alloc((size_t)M * 256); float* Y = ara.alloc((size_t)M * 256); float* O = ara.alloc((size_t)M * 256);
    float* W = ara.alloc(256 * 256); float* b = ara.alloc(256); float* rbv = ara.alloc(4 * 256); float* sl = ara.alloc(4);
    int* rv = reinterpret_cast<int*>(ara.alloc((size_t)M / L + 4));
    unsigned short* ws = reinterpret_cast<unsigned short*>(ara.alloc(8 * 8192 + 4));
    unsigned short* p8 = reinterpret_cast<unsigned short*>(ara.alloc((size_t)256 * 256 + 4));
    RAMP_REQUIRE(Q && R && Y && O && W && b && rbv && sl && rv && ws && p8, "hipMalloc failed");
    auto fill = [&](float* p, size_t n, unsigned seed, float sc) { hipLaunchKernelGGL(fill_uniform_kernel, dim3(2048), dim3(256), 0, sa, p, (long)n, seed, sc); };
    fill(Q, (size_t)M * 768, 1u, 1.5f); fill(R, (size_t)M * 256, 4u, 1.f); fill(W, 256 * 256, 2u, 1.f / 16.f); fill(b, 256, 5u, 1.f); fill(rbv, 1024, 6u, 1.f);
    const int pat[2] = {0, 1};
    int* dpat = reinterpret_cast<int*>(ara.alloc(4));
    RAMP_REQUIRE(dpat, "hipMalloc failed");
    RAMP_HIP_CHECK(hipMemcpyAsync(dpat, pat, sizeof(pat), hipMemcpyHostToDevice, sa));
    hipLaunchKernelGGL(fill_pattern_kernel, dim3(64), dim3(256), 0, sa, rv, dpat, 2, M / L);
    CK(init_atk_attributes());
    CK(ato_pack(W, 16384.f, ws, sa));
    CK(launch_pack_h3(W, p8, 256, 256, 16384.f, sa));
    const float one[4] = {1.5f, 0.f, 0.f, 0.f};
    RAMP_HIP_CHECK(hipMemcpyAsync(sl, one, sizeof(one), hipMemcpyHostToDevice, sa));
    AtoArgs a; a.M = M; a.L = L; a.QKV = Q; a.W = ws; a.bias = b; a.resid = R; a.Y = Y; a.amax_in = sl; a.amax_out = sl + 1; a.wsi = 1.f / 16384.f;
    a.range_flag = reinterpret_cast<int*>(sl + 2);
    if (flags & 1) { a.rowbias = rbv; a.rowvar = rv; a.rb_stride = 256; a.n_var = 2; }
    TklArgs t; t.M = M; t.N = 256; t.X = O; t.Y = Y; t.ldy = 256; t.W = p8; t.bias = b; t.resid = R; t.ldr = 256; t.amax_in = sl; t.amax_out = sl + 1; t.wsi = 1.f / 16384.f;
    t.range_flag = reinterpret_cast<int*>(sl + 2);
    if (flags & 1) { t.rowbias = rbv; t.rowvar = rv; t.rb_stride = 256; t.L = L; t.n_var = 2; }
    auto go = [&]() -> int {
      if (mode == 10) return launch_ato(a, sa);
      if (int rc = launch_attn_fwd(Q, O, M / L, L, sa)) return rc;
      return launch_tkl(t, sa);
    };
    unsigned long long* stamps = nullptr;
    if (flags & 256) {                                   // the stamped twin: where a head step's cycles go
      stamps = reinterpret_cast<unsigned long long*>(ara.alloc(256 * 4 * 8 * 2));
      RAMP_REQUIRE(stamps && mode == 10 && (flags & 1), "stamps: mode 10 with flags & 1");
      RAMP_HIP_CHECK(hipMemsetAsync(stamps, 0, 256 * 4 * 8 * 8, sa));
      a.stamps = stamps; a.ablate = (flags >> 9) & 7;
    }
    for (int i = 0; i < warmup; ++i) CK(go());
    hipEvent_t e0, e1;
    RAMP_HIP_CHECK(hipEventCreate(&e0)); RAMP_HIP_CHECK(hipEventCreate(&e1));
    RAMP_HIP_CHECK(hipEventRecord(e0, sa));
    int rca = 0;
    for (int i = 0; i < iters && rca == 0; ++i) { rca = go(); if (rca == 0) STRESS(Y, (size_t)M * 256, sa); }
    RAMP_HIP_CHECK(hipEventRecord(e1, sa));
    RAMP_HIP_CHECK(hipEventSynchronize(e1));
    float msa = 0.f;
    RAMP_HIP_CHECK(hipEventElapsedTime(&msa, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_us = msa * 1e3f / iters;
    if (rca == 0 && stamps) {
      std::vector<unsigned long long> hst(256 * 4 * 8);
      RAMP_HIP_CHECK(hipMemcpy(hst.data(), stamps, hst.size() * 8, hipMemcpyDeviceToHost));
      double sm[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int nw = 0;
      for (int w = 0; w < 1024; ++w) if (hst[w * 8 + 6]) { ++nw; for (int j = 0; j < 8; ++j) sm[j] += (double)hst[w * 8 + j]; }
      const int T4 = 192, n_tiles = (M + T4 - 1) / T4;
      const double steps = std::max(1, nw) * 4.0 * (double)((n_tiles + 255) / 256);
      fprintf(stderr, "[ato stamps] per head step (s_memtime ticks, %d waves): epilogue + loop top %.0f, head-start wait %.0f, S^T + softmax %.0f, PV %.0f, "
              "barriers %.0f, DMA issue + slab bodies %.0f; whole kernel %.0f per step; shader clock %.0f MHz\n",
              nw, sm[0] / steps, sm[1] / steps, sm[2] / steps, sm[3] / steps, sm[4] / steps, sm[5] / steps, sm[6] / steps, sm[7] > 0 ? sm[6] / sm[7] * 100.0 : 0.0);
    }
    return rca;
  }
  if (mode == 6 || mode == 7) {                        // ffx.hip: fused feed-forward with token-owning waves, forward / backward
    hipStream_t s6 = as_stream(stream);
    DevArena ar6;
    const size_t mt = ((size_t)M + 127) / 128;
    float* z1 = ar6.alloc((size_t)M * 256); float* dz = ar6.alloc((size_t)M * 256); float* out = ar6.alloc((size_t)M * 256);
    float* W1 = ar6.alloc(2048 * 256); float* W2 = ar6.alloc(256 * 1024); float* b1 = ar6.alloc(2048); float* b2 = ar6.alloc(256);
    float* lg = ar6.alloc(256); float* lb = ar6.alloc(256); float* stash = ar6.alloc(mt * 128 * 2048); float* sl = ar6.alloc(12);
    RAMP_REQUIRE(z1 && dz && out && W1 && W2 && b1 && b2 && lg && lb && stash && sl, "hipMalloc failed");
    auto fill6 = [&](float* p, size_t n, unsigned seed, float sc) { hipLaunchKernelGGL(fill_uniform_kernel, dim3(2048), dim3(256), 0, s6, p, (long)n, seed, sc); };
    fill6(z1, (size_t)M * 256, 1u, 1.f); fill6(dz, (size_t)M * 256, 7u, 1.f); fill6(W1, 2048 * 256, 2u, 1.f / 16.f); fill6(W2, 256 * 1024, 3u, 1.f / 32.f);
    fill6(b1, 2048, 5u, 1.f); fill6(b2, 256, 6u, 1.f); fill6(lg, 256, 8u, 1.f); fill6(lb, 256, 9u, 1.f);
    FfxPack pk;
    const bool s16 = (flags >> 16) & 1;                  // flags bit 16: the v_mfma_f32_16x16x32_f16 pair (ffx16.hip)
    CK(ffx_pack_all(ar6, W1, b1, W2, &pk, s6, s16));
    auto launch_ffx = [s16](const FfxArgs& a, bool bwd, hipStream_t st) { return s16 ? ramp::launch_ffx16(a, bwd, st) : ramp::launch_ffx(a, bwd, st); };
    const float one[12] = {4.f, 2.f, 1.f, 1.f, 0, 0, 0, 0, 0, 0, 0, 0};
    RAMP_HIP_CHECK(hipMemcpyAsync(sl, one, sizeof(one), hipMemcpyHostToDevice, s6));
    FfxArgs f; f.M = M; f.X = z1; f.Z1 = z1; f.Y = out; f.stash = stash; f.ln_g = lg; f.ln_b = lb; f.Wstream = pk.stream_f; f.b1 = pk.b1_pk; f.b2 = b2;
    f.amax_in1 = sl; f.amax_out1 = sl + 4; f.wsi1 = pk.wsi_w1; f.amax_in2 = sl + 1; f.amax_out2 = sl + 5; f.wsi2 = pk.wsi_w2; f.site2 = 1;
    f.range_flag = reinterpret_cast<int*>(sl + 8); f.ablate = (flags >> 8) & 255;
    f.half_mode = s16 ? (flags >> 17) & 3 : 0;          // flags bits 17-18 (ffx16.hip): 1 full tiles only, 2 half tiles only (0: launch_ffx16's own policy)
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(ar6.alloc(256 * 4 * 6 * 2));
    if (f.ablate & 64) { RAMP_REQUIRE(stamps, "hipMalloc failed"); RAMP_HIP_CHECK(hipMemsetAsync(stamps, 0, 256 * 4 * 6 * 8, s6)); f.stamps = stamps; }
    FfxArgs g = f; g.X = dz; g.Wstream = pk.stream_b; g.amax_in1 = sl + 2; g.amax_out1 = sl + 6; g.wsi1 = pk.wsi_w2; g.amax_in2 = sl + 3; g.amax_out2 = sl + 7; g.wsi2 = pk.wsi_w1;
    CK(launch_ffx(f, false, s6));                        // the backward kernel reads this stash
    auto go6 = [&]() { return mode == 6 ? launch_ffx(f, false, s6) : launch_ffx(g, true, s6); };
    for (int i = 0; i < warmup; ++i) CK(go6());
    hipEvent_t e0, e1;
    RAMP_HIP_CHECK(hipEventCreate(&e0)); RAMP_HIP_CHECK(hipEventCreate(&e1));
    RAMP_HIP_CHECK(hipEventRecord(e0, s6));
    int rc6 = 0;
    for (int i = 0; i < iters && rc6 == 0; ++i) { rc6 = go6(); if (rc6 == 0) STRESS(out, (size_t)M * 256, s6); }
    RAMP_HIP_CHECK(hipEventRecord(e1, s6));
    RAMP_HIP_CHECK(hipEventSynchronize(e1));
    float ms6 = 0.f;
    RAMP_HIP_CHECK(hipEventElapsedTime(&ms6, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_us = ms6 * 1e3f / iters;
    if (rc6 == 0 && (f.ablate & 64)) {                   // per-wave cycle sums of the LAST launch, averaged, on stderr
      std::vector<unsigned long long> h(256 * 4 * 6);
      RAMP_HIP_CHECK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
      double sm[6] = {0, 0, 0, 0, 0, 0}; int nw = 0;
      for (int w = 0; w < 1024; ++w) if (h[w * 6 + 5]) { ++nw; for (int j = 0; j < 6; ++j) sm[j] += (double)h[w * 6 + j]; }
      const double slabs = std::max(1, nw) * 96.0 * (double)((mt + 255) / 256);
      fprintf(stderr, "[ffx stamps] per slab (s_memtime ticks): vm wait %.0f, barrier %.0f, DMA issue %.0f, body %.0f (%d waves); shader clock %.0f MHz\n",
              sm[0] / slabs, sm[1] / slabs, sm[2] / slabs, sm[3] / slabs, nw, sm[5] > 0 ? sm[4] / sm[5] * 100.0 : 0.0);
    }
    return rc6;
  }
  if (mode == 9) {                                     // tkl.hip, tklb_kernel: d(ln1) + LayerNorm-1 backward (N, K ignored: 768 -> 256)
    hipStream_t s9 = as_stream(stream);
    DevArena ar9;
    float* X9 = ar9.alloc((size_t)M * 768); float* Z9 = ar9.alloc((size_t)M * 256); float* A9 = ar9.alloc((size_t)M * 256); float* Y9 = ar9.alloc((size_t)M * 256);
    float* W9 = ar9.alloc((size_t)256 * 768); float* lg = ar9.alloc(256); float* sl = ar9.alloc(4);
    unsigned short* p9 = reinterpret_cast<unsigned short*>(ar9.alloc((size_t)256 * 768 + 4));
    RAMP_REQUIRE(X9 && Z9 && A9 && Y9 && W9 && lg && sl && p9, "hipMalloc failed");
    auto fill9 = [&](float* p, size_t n, unsigned seed, float sc) { hipLaunchKernelGGL(fill_uniform_kernel, dim3(2048), dim3(256), 0, s9, p, (long)n, seed, sc); };
    fill9(X9, (size_t)M * 768, 1u, 1.f); fill9(Z9, (size_t)M * 256, 3u, 1.f); fill9(A9, (size_t)M * 256, 4u, 1.f); fill9(W9, (size_t)256 * 768, 2u, 1.f / 16.f); fill9(lg, 256, 8u, 1.f);
    CK(launch_pack_h3(W9, p9, 256, 768, 16384.f, s9));
    const float one[4] = {1.f, 0.f, 0.f, 0.f};
    RAMP_HIP_CHECK(hipMemcpyAsync(sl, one, sizeof(one), hipMemcpyHostToDevice, s9));
    TklbArgs a; a.M = M; a.X = X9; a.Z = Z9; a.add = A9; a.Y = Y9; a.W = p9; a.ln_g = lg; a.amax_in = sl; a.amax_out = sl + 1; a.wsi = 1.f / 16384.f;
    a.range_flag = reinterpret_cast<int*>(sl + 2);
    for (int i = 0; i < warmup; ++i) CK(launch_tklb(a, s9));
    hipEvent_t e0, e1;
    RAMP_HIP_CHECK(hipEventCreate(&e0)); RAMP_HIP_CHECK(hipEventCreate(&e1));
    RAMP_HIP_CHECK(hipEventRecord(e0, s9));
    int rc9 = 0;
    for (int i = 0; i < iters && rc9 == 0; ++i) { rc9 = launch_tklb(a, s9); if (rc9 == 0) STRESS(Y9, (size_t)M * 256, s9); }
    RAMP_HIP_CHECK(hipEventRecord(e1, s9));
    RAMP_HIP_CHECK(hipEventSynchronize(e1));
    float ms9 = 0.f;
    RAMP_HIP_CHECK(hipEventElapsedTime(&ms9, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_us = ms9 * 1e3f / iters;
    return rc9;
  }
  if (mode == 8) {                                     // tkl.hip: token-owning linear, K = 256; flags: 1 LayerNorm first, 2 bias + residual, >> 8 ablation
    hipStream_t s8 = as_stream(stream);
    DevArena ar8;
    RAMP_REQUIRE(K == 256 && N % 32 == 0 && N <= 768, "mode 8: K = 256, N a multiple of 32 up to 768");
    float* X8 = ar8.alloc((size_t)M * 256); float* Y8 = ar8.alloc((size_t)M * N); float* R8 = ar8.alloc((size_t)M * N);
    float* W8 = ar8.alloc((size_t)N * 256); float* b8 = ar8.alloc(N); float* lg = ar8.alloc(256); float* lb = ar8.alloc(256); float* sl = ar8.alloc(4);
    unsigned short* p8 = reinterpret_cast<unsigned short*>(ar8.alloc((size_t)N * 256 + 4));
    RAMP_REQUIRE(X8 && Y8 && R8 && W8 && b8 && lg && lb && sl && p8, "hipMalloc failed");
    auto fill8 = [&](float* p, size_t n, unsigned seed, float sc) { hipLaunchKernelGGL(fill_uniform_kernel, dim3(2048), dim3(256), 0, s8, p, (long)n, seed, sc); };
    fill8(X8, (size_t)M * 256, 1u, 1.f); fill8(R8, (size_t)M * N, 4u, 1.f); fill8(W8, (size_t)N * 256, 2u, 1.f / 16.f); fill8(b8, N, 5u, 1.f);
    fill8(lg, 256, 8u, 1.f); fill8(lb, 256, 9u, 1.f);
    const bool s16 = (flags >> 16) & 1;                  // flags bit 16: the v_mfma_f32_16x16x32_f16 kernel (tkl16.hip)
    if (s16) { float* tmp8 = ar8.alloc((size_t)N * 256); RAMP_REQUIRE(tmp8, "hipMalloc failed"); CK(ffx16_pack(W8, N, 256, 0, 16384.f, tmp8, p8, s8)); }
    else CK(launch_pack_h3(W8, p8, N, 256, 16384.f, s8));
    auto launch_tkl = [s16](const TklArgs& t, hipStream_t st) { return s16 ? ramp::launch_tkl16(t, st) : ramp::launch_tkl(t, st); };
    const float one[4] = {(flags & 1) ? 4.f : 1.f, 0.f, 0.f, 0.f};
    RAMP_HIP_CHECK(hipMemcpyAsync(sl, one, sizeof(one), hipMemcpyHostToDevice, s8));
    TklArgs a; a.M = M; a.N = N; a.X = X8; a.Y = Y8; a.ldy = N; a.W = p8; a.amax_in = sl; a.amax_out = sl + 1; a.wsi = 1.f / 16384.f;
    a.range_flag = reinterpret_cast<int*>(sl + 2); a.ablate = (flags >> 8) & 255;
    if (flags & 1) { a.ln_g = lg; a.ln_b = lb; }
    if (flags & 2) { a.bias = b8; a.resid = R8; a.ldr = N; }
    for (int i = 0; i < warmup; ++i) CK(launch_tkl(a, s8));
    hipEvent_t e0, e1;
    RAMP_HIP_CHECK(hipEventCreate(&e0)); RAMP_HIP_CHECK(hipEventCreate(&e1));
    RAMP_HIP_CHECK(hipEventRecord(e0, s8));
    int rc8 = 0;
    for (int i = 0; i < iters && rc8 == 0; ++i) { rc8 = launch_tkl(a, s8); if (rc8 == 0) STRESS(Y8, (size_t)M * N, s8); }
    RAMP_HIP_CHECK(hipEventRecord(e1, s8));
    RAMP_HIP_CHECK(hipEventSynchronize(e1));
    float ms8 = 0.f;
    RAMP_HIP_CHECK(hipEventElapsedTime(&ms8, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_us = ms8 * 1e3f / iters;
    return rc8;
  }
  if (mode == 5) {                                     // the fused FF1 -> GEGLU -> FF2 kernel (N, K ignored: 256 -> 2 x 1024 -> 256)
    hipStream_t s5 = as_stream(stream);
    DevArena ar5;
    float* A5 = ar5.alloc((size_t)M * 256); float* W1 = ar5.alloc(2048 * 256); float* W2 = ar5.alloc(256 * 1024);
    float* st5 = ar5.alloc((size_t)M * 2048); float* z1 = ar5.alloc((size_t)M * 256); float* z2 = ar5.alloc((size_t)M * 256);
    float* b1 = ar5.alloc(2048); float* b2 = ar5.alloc(256); float* sl = ar5.alloc(8);
    unsigned short* p1 = reinterpret_cast<unsigned short*>(ar5.alloc(2048 * 256 + 4));
    unsigned short* p2 = reinterpret_cast<unsigned short*>(ar5.alloc(256 * 1024 + 4));
    RAMP_REQUIRE(A5 && W1 && W2 && st5 && z1 && z2 && b1 && b2 && sl && p1 && p2, "hipMalloc failed");
    auto fill5 = [&](float* p, size_t n, unsigned seed, float sc) { hipLaunchKernelGGL(fill_uniform_kernel, dim3(2048), dim3(256), 0, s5, p, (long)n, seed, sc); };
    fill5(A5, (size_t)M * 256, 1u, 1.f); fill5(W1, 2048 * 256, 2u, 1.f / 16.f); fill5(W2, 256 * 1024, 3u, 1.f / 32.f);
    fill5(z1, (size_t)M * 256, 4u, 1.f); fill5(b1, 2048, 5u, 1.f); fill5(b2, 256, 6u, 1.f);
    const float one[8] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f};
    RAMP_HIP_CHECK(hipMemcpyAsync(sl, one, 32, hipMemcpyHostToDevice, s5));
    CK(launch_pack_h3(W1, p1, 2048, 256, 16384.f, s5)); CK(launch_pack_h3(W2, p2, 256, 1024, 32768.f, s5));
    GemmArgs g1; g1.A = A5; g1.lda = 256; g1.W = W1; g1.Wx = p1; g1.wx_packed = 2; g1.w_scale_inv = 1.f / 16384.f; g1.bias = b1;
    g1.C = st5; g1.ldc = 2048; g1.M = M; g1.N = 2048; g1.K = 256; g1.epi = EPI_GEGLU_FWD; g1.geglu_group = 32;
    g1.a_absmax_in = sl; g1.a_absmax_out = sl + 1; g1.range_flag = reinterpret_cast<int*>(sl + 2); g1.ablate = (flags >> 8) & 31;
    GemmArgs g2; g2.W = W2; g2.Wx = p2; g2.wx_packed = 2; g2.w_scale_inv = 1.f / 32768.f; g2.bias = b2; g2.resid = z1; g2.ldr = 256;
    g2.C = z2; g2.ldc = 256; g2.M = M; g2.N = 256; g2.K = 1024;
    g2.a_absmax_in = sl + 4; g2.a_absmax_out = sl + 5; g2.range_flag = reinterpret_cast<int*>(sl + 6); g2.site_id = 1;
    for (int i = 0; i < warmup; ++i) CK(launch_ff_fwd(g1, g2, s5));
    hipEvent_t e0, e1;
    RAMP_HIP_CHECK(hipEventCreate(&e0)); RAMP_HIP_CHECK(hipEventCreate(&e1));
    RAMP_HIP_CHECK(hipEventRecord(e0, s5));
    int rc5 = 0;
    for (int i = 0; i < iters && rc5 == 0; ++i) { rc5 = launch_ff_fwd(g1, g2, s5); if (rc5 == 0) STRESS(z2, (size_t)M * 256, s5); }
    RAMP_HIP_CHECK(hipEventRecord(e1, s5));
    RAMP_HIP_CHECK(hipEventSynchronize(e1));
    float ms5 = 0.f;
    RAMP_HIP_CHECK(hipEventElapsedTime(&ms5, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_us = ms5 * 1e3f / iters;
    return rc5;
  }
  hipStream_t s = as_stream(stream);
  DevArena ar;
  auto fill = [&](float* p, size_t n, unsigned seed, float sc) {
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(2048), dim3(256), 0, s, p, (long)n, seed, sc);
  };
  const bool geglu = flags & 4, amul = flags & 8;
  const int Ka = amul ? K / 2 : K;
  float* A = ar.alloc((size_t)M * Ka); float* W = ar.alloc((size_t)taps * N * K); float* C = ar.alloc((size_t)M * N);
  float* bias = ar.alloc(N); float* R = (flags & 2) ? ar.alloc((size_t)M * N) : nullptr;
  float* aux = geglu ? ar.alloc((size_t)M * N / 2) : nullptr; float* mul = amul ? ar.alloc((size_t)M * K) : nullptr;
  float* slots = ar.alloc(4);
  RAMP_REQUIRE(A && W && C && bias && slots && (!(flags & 2) || R) && (!geglu || aux) && (!amul || mul), "hipMalloc failed");
  fill(A, (size_t)M * Ka, 1u, 1.f); fill(W, (size_t)taps * N * K, 2u, 1.f / std::sqrt((float)K * taps)); fill(bias, N, 3u, 1.f);
  if (R) fill(R, (size_t)M * N, 4u, 1.f);
  if (mul) fill(mul, (size_t)M * K, 5u, 1.f);
  RAMP_HIP_CHECK(hipMemsetAsync(slots, 0, 16, s));
  const float one = 1.f;
  RAMP_HIP_CHECK(hipMemcpyAsync(slots, &one, 4, hipMemcpyHostToDevice, s));
  GemmArgs a; a.A = A; a.lda = Ka; a.W = W; a.bias = (flags & 1) ? bias : nullptr; a.resid = R; a.ldr = N; a.C = C; a.ldc = N;
  a.M = M; a.N = N; a.K = K; a.taps = taps; a.L = L;
  if (taps > 1) { a.shift0 = -(taps / 2); a.shift_step = 1; }
  if (geglu) { a.epi = EPI_GEGLU_FWD; a.aux_out = aux; a.ld_aux = N / 2; a.geglu_group = (mode == 1 || mode == 3) ? 32 : 64; }
  if (amul) { a.Amul = mul; a.lda_mul = K; a.a_period = Ka; }
  a.tile_pref = (flags & 16) ? 1 : (flags & 32) ? 3 : 0;
  a.ablate = ((flags >> 8) & 255) | ((flags & (1 << 30)) ? 256 : 0);
  const long n = (long)taps * N * K;
  const bool frag_ok = N >= 64 && N % 32 == 0 && K % 16 == 0;
  unsigned short* planes = nullptr;
  RAMP_REQUIRE(mode != 4, "mode 4 (the experimental LDS-DMA tile GEMM) was removed from the library in round 5");
  if (mode == 3 && frag_ok) {
    planes = reinterpret_cast<unsigned short*>(ar.alloc((size_t)n + 4));
    RAMP_REQUIRE(planes, "hipMalloc failed");
    const float sc = std::ldexp(1.f, 10) * std::sqrt((float)K * taps);      // max |w| ~ 1 / sqrt(K taps)
    float scp = 1.f; { int e; std::frexp(sc, &e); scp = std::ldexp(1.f, e - 1); }
    CK(launch_pack_h3(W, planes, (long)taps * N, K, scp, s));
    a.Wx = planes; a.wx_packed = 2; a.w_scale_inv = 1.f / scp;
    a.a_absmax_in = slots; a.a_absmax_out = slots + 1; a.range_flag = reinterpret_cast<int*>(slots + 2);
  } else if (mode == 1 && frag_ok) {
    planes = reinterpret_cast<unsigned short*>(ar.alloc((3 * (size_t)n + 1) / 2 + 4));
    RAMP_REQUIRE(planes, "hipMalloc failed");
    CK(launch_pack_x6(W, planes, (long)taps * N, K, s));
    a.Wx = planes; a.wx_packed = 1;
  } else if ((mode == 1 || mode == 2) && N >= 128) {
    planes = reinterpret_cast<unsigned short*>(ar.alloc((3 * (size_t)n + 1) / 2 + 4));
    RAMP_REQUIRE(planes, "hipMalloc failed");
    CK(launch_split3(W, planes, n, s));
    a.Wx = planes; a.wx_plane = n;
  }
  auto go = [&]() { return launch_gemm(a, s); };
  for (int i = 0; i < warmup; ++i) CK(go());
  hipEvent_t e0, e1;
  RAMP_HIP_CHECK(hipEventCreate(&e0)); RAMP_HIP_CHECK(hipEventCreate(&e1));
  RAMP_HIP_CHECK(hipEventRecord(e0, s));
  int rc = 0;
  for (int i = 0; i < iters && rc == 0; ++i) { rc = go(); if (rc == 0) STRESS(C, (size_t)M * N, s); }
  RAMP_HIP_CHECK(hipEventRecord(e1, s));
  RAMP_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  RAMP_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *avg_us = ms * 1e3f / iters;
  return rc;
}

int ramp_stress_gemm(int32_t M, int32_t N, int32_t K, int32_t taps, int32_t L, int32_t mode, int32_t flags, int32_t iters,
                     int64_t* mismatching_words, float* rel_err_vs_fp32, void* stream) {
  RAMP_REQUIRE(mismatching_words && iters >= 2, "bad arguments");
  StressHook hook;
  float us = 0.f;
  g_stress = &hook;
  int rc = 0;
  if (rel_err_vs_fp32 && mode >= 1 && mode <= 3) {     // the same operands (seeded fills) on the exact-fp32 MFMA kernel first
    hook.capture_ref = true;
    rc = ramp_bench_gemm(M, N, K, taps, L, 0, flags & 0xff, 0, 1, &us, stream);
    hook.capture_ref = false;
  }
  if (rc == 0) rc = ramp_bench_gemm(M, N, K, taps, L, mode, flags, 0, iters, &us, stream);
  g_stress = nullptr;
  if (rc != 0) return rc;
  unsigned long long h[2] = {0, 0};
  RAMP_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
  if (hook.mism) RAMP_HIP_CHECK(hipMemcpy(h, hook.mism, 16, hipMemcpyDeviceToHost));
  RAMP_REQUIRE(hook.launches == iters, "stress hook did not see every launch");
  *mismatching_words = (int64_t)h[0];
  if (rel_err_vs_fp32) {
    const unsigned e = (unsigned)(h[1] & 0xffffffffu), r = (unsigned)(h[1] >> 32);
    const float ef = __builtin_bit_cast(float, e), rf = __builtin_bit_cast(float, r);
    *rel_err_vs_fp32 = (hook.ref && rf > 0.f) ? ef / rf : -1.f;
  }
  return 0;
}


}  // extern "C"

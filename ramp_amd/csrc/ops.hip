// The context-free kernel-level entry points of the C ABI (include/ramp_hip.h): ramp_apf ... ramp_ddim_finish (what the Python mirror of the
// reference's helper functions calls one kernel at a time) and the ramp_op_* unit entry points the parity tests drive every kernel family through
// (each packs its weights from raw fp32 exactly as ramp_finalize_weights does).  Split from engine.hip in round 6: an edit of a kernel family's
// argument struct rebuilds this file and its own, not the sampler.
#include "engine_util.h"

extern "C" {

// ---- kernel-level entry points ---------------------------------------------------------------------
// The context-free entry points take small HOST arrays (window weights, waypoint indices).  They are staged through a
// per-thread ring of device slots allocated once, so a call neither allocates nor synchronises; a slot is reused after
// RING calls, by which time the stream-ordered kernel that read it has long been submitted behind 63 others.
namespace {
struct HostArgRing {
  static constexpr int RING = 64, SLOT = 1024;       // bytes per slot: 129 window weights or 256 indices
  // one ring per device (a thread that alternates devices keeps both); every slot carries the event recorded behind the
  // kernel that reads it, on whatever stream that was: before a slot is reused the event is waited for, so calls on
  // different streams cannot overwrite an array an earlier kernel has not read yet (normally complete long ago: 63 calls)
  struct PerDevice { char* base = nullptr; int next = 0; hipEvent_t ev[RING] = {}; bool used[RING] = {}; };
  std::map<int, PerDevice> rings;
  int cur_dev = -1, cur_slot = -1;
  int stage(const void* host, size_t bytes, hipStream_t s, void** out) {
    RAMP_REQUIRE(bytes <= (size_t)SLOT, "host argument array too long");
    int dev = 0; RAMP_HIP_CHECK(hipGetDevice(&dev));
    PerDevice& r = rings[dev];
    if (!r.base) RAMP_HIP_CHECK(hipMalloc(&r.base, (size_t)RING * SLOT));
    const int slot = r.next++ % RING;
    if (r.used[slot]) RAMP_HIP_CHECK(hipEventSynchronize(r.ev[slot]));
    else { RAMP_HIP_CHECK(hipEventCreateWithFlags(&r.ev[slot], hipEventDisableTiming)); r.used[slot] = true; }
    char* p = r.base + (size_t)slot * SLOT;
    RAMP_HIP_CHECK(hipMemcpyAsync(p, host, bytes, hipMemcpyHostToDevice, s));
    *out = p; cur_dev = dev; cur_slot = slot;
    return 0;
  }
  // after the kernel that reads the staged array has been launched on `s`
  int done(hipStream_t s) {
    if (cur_slot >= 0) RAMP_HIP_CHECK(hipEventRecord(rings[cur_dev].ev[cur_slot], s));
    cur_slot = -1;
    return 0;
  }
};
thread_local HostArgRing g_ring;
}  // namespace

int ramp_apf(float* traj, int32_t B, int32_t H, int32_t S, const ramp_apf_params* p, void* stream) {
  RAMP_REQUIRE(traj && p && p->cloud && p->window_weights_host, "null argument");
  RAMP_REQUIRE(p->window >= 0 && p->window <= 64, "bad window");
  hipStream_t s = as_stream(stream);
  void* w = nullptr;
  CK(g_ring.stage(p->window_weights_host, (2 * p->window + 1) * 4, s, &w));
  ApfArgs a; a.traj = traj; a.cloud = p->cloud; a.window = static_cast<const float*>(w); a.B = B; a.H = H; a.S = S;
  a.P = p->n_points; a.win = p->window; a.thr = p->threshold; a.strength = p->strength;
  for (int q = 0; q < std::max(1, p->passes); ++q) CK(launch_apf(a, s));
  return g_ring.done(s);
}

int ramp_apf_dynamic(float* traj, int32_t B, int32_t H, int32_t S, const double* points, int32_t n_points,
                     double thr_query, double thr_force, double strength, int32_t window, int32_t affected,
                     const float* goal, const int32_t* enable, void* stream) {
  RAMP_REQUIRE(traj && points, "null argument");
  ApfDynArgs a; a.traj = traj; a.points = points; a.goal = goal; a.enable = enable; a.B = B; a.H = H; a.S = S;
  a.P = n_points; a.window = window; a.affected = affected; a.thr_query = thr_query; a.thr_force = thr_force;
  a.strength = strength;
  return launch_apf_dynamic(a, as_stream(stream));
}

int ramp_hard_cond(float* x, int32_t B, int32_t H, int32_t S, int32_t n, const int32_t* idx_host, const float* val,
                   void* stream) {
  RAMP_REQUIRE(x && (n == 0 || (idx_host && val)), "null argument");
  if (n == 0) return 0;
  RAMP_REQUIRE(n <= 256, "too many hard conditions");
  for (int j = 0; j < n; ++j) RAMP_REQUIRE(idx_host[j] >= 0 && idx_host[j] < H, "hard index out of range");
  hipStream_t s = as_stream(stream);
  void* d = nullptr;
  CK(g_ring.stage(idx_host, (size_t)n * 4, s, &d));
  HardConds hc; hc.idx = static_cast<const int*>(d); hc.val = val; hc.n = n;
  CK(launch_hard_cond(x, hc, B, H, S, s));
  return g_ring.done(s);
}

int ramp_traj_costs(const float* traj, int32_t B, int32_t H, int32_t S, const float* cloud, int32_t n_points,
                    float threshold, int32_t* mask, float* path_len, float* smooth, void* stream) {
  RAMP_REQUIRE(traj && cloud && mask && path_len && smooth, "null argument");
  return launch_traj_costs(traj, cloud, B, H, S, n_points, threshold, mask, path_len, smooth, as_stream(stream));
}
int ramp_traj_metrics(const float* traj, int32_t B, int32_t H, int32_t S, const float* box_centers, const float* box_sizes,
                      int32_t n_boxes, float* intensity, float* path_len, float* smooth, void* stream) {
  RAMP_REQUIRE(traj && intensity && path_len && smooth && (n_boxes == 0 || (box_centers && box_sizes)), "null argument");
  return launch_traj_metrics(traj, B, H, S, box_centers, box_sizes, n_boxes, intensity, path_len, smooth, as_stream(stream));
}
int ramp_waypoint_variance(const float* traj, int32_t B, int32_t H, int32_t S, double* scratch, double* out, void* stream) {
  RAMP_REQUIRE(traj && scratch && out, "null argument");
  return launch_waypoint_variance(traj, B, H, S, scratch, out, as_stream(stream));
}

int ramp_cfg_mean(const float* x, const float* eps, int32_t B, int32_t HS, int32_t n_rp, double w0, double w1,
                  float sqrt_recip, float sqrt_recipm1, float coef1, float coef2, int32_t clip, int32_t predict_x0, float* x0_out,
                  float* mean_out, float* ecomb_out, void* stream) {
  RAMP_REQUIRE(x && eps, "null argument");
  CfgMeanArgs m; m.x = x; m.eps = eps; m.B = B; m.HS = HS; m.n_rp = n_rp; m.w0 = (float)w0; m.w1 = (float)w1;
  m.w0p1 = (float)(1.0 + w0); m.sqrt_recip = sqrt_recip; m.sqrt_recipm1 = sqrt_recipm1; m.coef1 = coef1; m.coef2 = coef2;
  m.clip = clip; m.predict_x0 = predict_x0 != 0; m.x0 = x0_out; m.mean = mean_out; m.ecomb = ecomb_out;
  return launch_cfg_mean(m, as_stream(stream));
}

int ramp_ddim_finish(const float* x, const float* x0, float sqrt_a_t, float sqrt_1m_a_t, float sqrt_a_prev,
                     float dir_coef, float* x_out, int32_t B, int32_t H, int32_t S, void* stream) {
  RAMP_REQUIRE(x && x0 && x_out, "null argument");
  HardConds hc;
  return launch_ddim_finish(x, x0, sqrt_a_t, sqrt_1m_a_t, sqrt_a_prev, dir_coef, hc, x_out, nullptr, B, H, S,
                            as_stream(stream));
}

int ramp_op_gemm(const float* A, const float* W, const float* bias, const float* resid, float* C, int32_t M, int32_t N,
                 int32_t K, int32_t taps, int32_t shift0, int32_t shift_step, int32_t L, void* stream) {
  return ramp_op_gemm_mode(A, W, bias, resid, C, M, N, K, taps, shift0, shift_step, L, 0, 0.f, nullptr, nullptr, stream);
}

int ramp_op_gemm_mode(const float* A, const float* W, const float* bias, const float* resid, float* C, int32_t M,
                      int32_t N, int32_t K, int32_t taps, int32_t shift0, int32_t shift_step, int32_t L, int32_t mode,
                      float a_absmax_prev, float* a_absmax_out_host, int32_t* range_flag_out_host, void* stream) {
  RAMP_REQUIRE(A && W && C, "null argument");
  RAMP_REQUIRE(mode >= 0 && mode <= 5 && mode != 4, "mode: 0 fp32, 1 bf16x6, 2 bf16x6 (LDS-staged weights), 3 fp16x3, 5 fp16x3 "
                                       "sample-owning k = 5 convolution (tkc.hip)");
  hipStream_t s = as_stream(stream);
  if (mode == 5) {
    RAMP_REQUIRE(taps == 5 && ((shift0 == -2 && shift_step == 1) || (shift0 == 2 && shift_step == -1)) && tkc_applicable(M, L, N, K, nullptr),
                 "mode 5: a k = 5 convolution (or its input gradient) with C_in, C_out in {32, 64}, L >= 8 dividing 48 or 32");
    DevArena ar;
    std::vector<float> hw((size_t)5 * N * K);
    RAMP_HIP_CHECK(hipMemcpy(hw.data(), W, hw.size() * 4, hipMemcpyDeviceToHost));
    float mx = 0.f; for (float v : hw) mx = std::max(mx, std::fabs(v));
    float sc = 1.f;
    if (mx > 0.f && std::isfinite(mx)) { int e; std::frexp(mx, &e); sc = std::ldexp(1.f, 11 - e); }
    unsigned short* pl = reinterpret_cast<unsigned short*>(ar.alloc(tkc_packed_halves(N, K) / 2 + 4));
    float* sl = ar.alloc(4);
    RAMP_REQUIRE(pl && sl, "hipMalloc failed");
    CK(init_tkc_attributes());
    CK(tkc_pack(W, N, K, sc, pl, s));
    const float v[4] = {a_absmax_prev, 0.f, 0.f, 0.f};
    RAMP_HIP_CHECK(hipMemcpyAsync(sl, v, 16, hipMemcpyHostToDevice, s));
    TkcArgs t; t.M = M; t.L = L; t.N = N; t.K = K; t.dir = shift_step; t.X = A; t.ldx = K; t.W = pl; t.bias = bias; t.resid = resid; t.ldr = N;
    t.Y = C; t.ldy = N; t.amax_in = a_absmax_prev > 0.f ? sl : nullptr; t.amax_out = sl + 1; t.wsi = 1.f / sc; t.range_flag = reinterpret_cast<int*>(sl + 2);
    int rc5 = launch_tkc(t, s);
    hipError_t e5 = hipStreamSynchronize(s);
    float back[4] = {0, 0, 0, 0};
    if (rc5 == 0 && e5 == hipSuccess) {
      e5 = hipMemcpy(back, sl, sizeof(back), hipMemcpyDeviceToHost);
      if (a_absmax_out_host) *a_absmax_out_host = back[1];
      if (range_flag_out_host) std::memcpy(range_flag_out_host, &back[2], 4);
    }
    RAMP_HIP_CHECK(e5);
    return rc5;
  }
  GemmArgs a; a.A = A; a.lda = K; a.W = W; a.bias = bias; a.resid = resid; a.ldr = N; a.C = C; a.ldc = N;
  a.M = M; a.N = N; a.K = K; a.taps = taps; a.shift0 = shift0; a.shift_step = shift_step; a.L = L;
  const long n = (long)taps * N * K;
  const bool frag_ok = N >= 64 && N % 32 == 0 && K % 16 == 0;
  unsigned short* planes = nullptr; float* slots = nullptr;
  int rc = 0;
  if (mode == 3 && frag_ok) {
    // the product's static weight scale: max |w| -> [2^10, 2^11)  (ramp_finalize_weights)
    std::vector<float> hw(n);
    RAMP_HIP_CHECK(hipMemcpy(hw.data(), W, n * sizeof(float), hipMemcpyDeviceToHost));
    float mx = 0.f; for (float v : hw) mx = std::max(mx, std::fabs(v));
    float sc = 1.f;
    if (mx > 0.f && std::isfinite(mx)) { int e; std::frexp(mx, &e); sc = std::ldexp(1.f, 11 - e); }
    RAMP_HIP_CHECK(hipMalloc(&planes, 2 * n * sizeof(unsigned short)));
    RAMP_HIP_CHECK(hipMalloc(&slots, 16));
    const float v[4] = {a_absmax_prev, 0.f, 0.f, 0.f};
    RAMP_HIP_CHECK(hipMemcpyAsync(slots, v, 16, hipMemcpyHostToDevice, s));
    rc = launch_pack_h3(W, planes, (long)taps * N, K, sc, s);
    a.Wx = planes; a.wx_packed = 2; a.w_scale_inv = 1.f / sc;
    a.a_absmax_in = a_absmax_prev > 0.f ? slots : nullptr; a.a_absmax_out = slots + 1;
    a.range_flag = reinterpret_cast<int*>(slots + 2);
  } else if (mode == 1 && frag_ok) {
    RAMP_HIP_CHECK(hipMalloc(&planes, 3 * n * sizeof(unsigned short)));
    rc = launch_pack_x6(W, planes, (long)taps * N, K, s);
    a.Wx = planes; a.wx_packed = 1;
  } else if ((mode == 1 || mode == 2) && N >= 128) {
    RAMP_HIP_CHECK(hipMalloc(&planes, 3 * n * sizeof(unsigned short)));
    rc = launch_split3(W, planes, n, s);
    a.Wx = planes; a.wx_plane = n;
  }
  if (rc == 0) rc = launch_gemm(a, s);
  hipError_t e = hipStreamSynchronize(s);
  if (rc == 0 && e == hipSuccess && slots) {
    float back[4] = {0, 0, 0, 0};
    e = hipMemcpy(back, slots, 16, hipMemcpyDeviceToHost);
    if (a_absmax_out_host) *a_absmax_out_host = back[1];
    int fl; std::memcpy(&fl, &back[2], 4);
    if (range_flag_out_host) *range_flag_out_host = fl;
  } else {
    if (a_absmax_out_host) *a_absmax_out_host = 0.f;
    if (range_flag_out_host) *range_flag_out_host = 0;
  }
  if (planes) (void)hipFree(planes);
  if (slots) (void)hipFree(slots);
  RAMP_HIP_CHECK(e);
  return rc;
}
static int op_ffx_impl(const float* z1, const float* dz, const float* W1, const float* b1, const float* W2, const float* b2,
                       const float* ln_g, const float* ln_b, int32_t M, const float* absmax_prev_host, float* z2, float* dz1,
                       float* absmax_out_host, int32_t* range_flag_out_host, void* stream, bool s16) {
  RAMP_REQUIRE(z1 && W1 && b1 && W2 && b2 && ln_g && ln_b && z2 && M > 0, "null argument");
  hipStream_t s = as_stream(stream);
  DevArena ar;
  FfxPack pk;
  CK(ffx_pack_all(ar, W1, b1, W2, &pk, s, s16));
  auto launch_ffx = [s16](const FfxArgs& a, bool bwd, hipStream_t st) { return s16 ? ramp::launch_ffx16(a, bwd, st) : ramp::launch_ffx(a, bwd, st); };
  const size_t mt = ((size_t)M + 127) / 128;
  float* stash = ar.alloc(mt * 128 * 2048); float* slots = ar.alloc(12);
  RAMP_REQUIRE(stash && slots, "hipMalloc failed");
  float host[12] = {0};
  for (int i = 0; i < 4; ++i) host[i] = absmax_prev_host ? absmax_prev_host[i] : 0.f;
  RAMP_HIP_CHECK(hipMemcpyAsync(slots, host, sizeof(host), hipMemcpyHostToDevice, s));
  FfxArgs f; f.M = M; f.X = z1; f.Z1 = z1; f.Y = z2; f.stash = stash; f.ln_g = ln_g; f.ln_b = ln_b; f.Wstream = pk.stream_f;
  f.b1 = pk.b1_pk; f.b2 = b2; f.range_flag = reinterpret_cast<int*>(slots + 8);
  f.amax_in1 = host[0] > 0.f ? slots + 0 : nullptr; f.amax_out1 = slots + 4; f.wsi1 = pk.wsi_w1; f.site1 = 0;
  f.amax_in2 = host[1] > 0.f ? slots + 1 : nullptr; f.amax_out2 = slots + 5; f.wsi2 = pk.wsi_w2; f.site2 = 1;
  int rc = launch_ffx(f, false, s);
  if (rc == 0 && dz && dz1) {
    FfxArgs g; g.M = M; g.X = dz; g.Z1 = z1; g.Y = dz1; g.stash = stash; g.ln_g = ln_g; g.ln_b = ln_b; g.Wstream = pk.stream_b;
    g.range_flag = reinterpret_cast<int*>(slots + 8);
    g.amax_in1 = host[2] > 0.f ? slots + 2 : nullptr; g.amax_out1 = slots + 6; g.wsi1 = pk.wsi_w2; g.site1 = 2;
    g.amax_in2 = host[3] > 0.f ? slots + 3 : nullptr; g.amax_out2 = slots + 7; g.wsi2 = pk.wsi_w1; g.site2 = 3;
    rc = launch_ffx(g, true, s);
  }
  hipError_t e = hipStreamSynchronize(s);
  if (rc == 0 && e == hipSuccess) {
    e = hipMemcpy(host, slots, sizeof(host), hipMemcpyDeviceToHost);
    if (absmax_out_host) for (int i = 0; i < 4; ++i) absmax_out_host[i] = host[4 + i];
    if (range_flag_out_host) std::memcpy(range_flag_out_host, &host[8], 4);
  }
  RAMP_HIP_CHECK(e);
  return rc;
}
int ramp_op_ffx(const float* z1, const float* dz, const float* W1, const float* b1, const float* W2, const float* b2,
                const float* ln_g, const float* ln_b, int32_t M, const float* absmax_prev_host, float* z2, float* dz1,
                float* absmax_out_host, int32_t* range_flag_out_host, void* stream) {
  return op_ffx_impl(z1, dz, W1, b1, W2, b2, ln_g, ln_b, M, absmax_prev_host, z2, dz1, absmax_out_host, range_flag_out_host, stream, false);
}
int ramp_op_ffx16(const float* z1, const float* dz, const float* W1, const float* b1, const float* W2, const float* b2,
                  const float* ln_g, const float* ln_b, int32_t M, const float* absmax_prev_host, float* z2, float* dz1,
                  float* absmax_out_host, int32_t* range_flag_out_host, void* stream) {
  return op_ffx_impl(z1, dz, W1, b1, W2, b2, ln_g, ln_b, M, absmax_prev_host, z2, dz1, absmax_out_host, range_flag_out_host, stream, true);
}

static int op_tkl_impl(const float* X, const float* W, const float* bias, const float* resid, const float* rowbias,
                       const int32_t* rowvar, int32_t n_var, int32_t L, const float* ln_g, const float* ln_b, int32_t M, int32_t N,
                       float absmax_prev, float* Y, float* absmax_out_host, int32_t* range_flag_out_host, void* stream, bool s16) {
  RAMP_REQUIRE(X && W && Y && M > 0 && N >= 32 && N % 32 == 0 && N <= 768, "bad arguments");
  hipStream_t s = as_stream(stream);
  DevArena ar;
  std::vector<float> hw((size_t)N * 256);
  RAMP_HIP_CHECK(hipMemcpy(hw.data(), W, hw.size() * 4, hipMemcpyDeviceToHost));
  float mx = 0.f;
  for (float v : hw) mx = std::max(mx, std::fabs(v));
  float sc = 1.f;
  if (mx > 0.f && std::isfinite(mx)) { int e; std::frexp(mx, &e); sc = std::ldexp(1.f, 11 - e); }
  unsigned short* planes = reinterpret_cast<unsigned short*>(ar.alloc((size_t)N * 256 + 4));
  float* slots = ar.alloc(4);
  RAMP_REQUIRE(planes && slots, "hipMalloc failed");
  if (s16) {      // 16 x 32 fragments (tkl16.hip)
    float* tmp = ar.alloc((size_t)N * 256);
    RAMP_REQUIRE(tmp, "hipMalloc failed");
    CK(ffx16_pack(W, N, 256, 0, sc, tmp, planes, s));
  } else CK(launch_pack_h3(W, planes, N, 256, sc, s));
  const float host[4] = {absmax_prev, 0.f, 0.f, 0.f};
  RAMP_HIP_CHECK(hipMemcpyAsync(slots, host, sizeof(host), hipMemcpyHostToDevice, s));
  TklArgs a; a.M = M; a.N = N; a.X = X; a.Y = Y; a.ldy = N; a.W = planes; a.bias = bias; a.resid = resid; a.ldr = N;
  a.rowbias = rowbias; a.rowvar = rowvar; a.row0 = 0; a.rb_stride = N; a.L = L > 0 ? L : 1; a.n_var = n_var;
  a.ln_g = ln_g; a.ln_b = ln_b; a.amax_in = absmax_prev > 0.f ? slots : nullptr; a.amax_out = slots + 1; a.wsi = 1.f / sc; a.site = 0;
  a.range_flag = reinterpret_cast<int*>(slots + 2);
  int rc = s16 ? launch_tkl16(a, s) : launch_tkl(a, s);
  hipError_t e = hipStreamSynchronize(s);
  float back[4] = {0, 0, 0, 0};
  if (rc == 0 && e == hipSuccess) {
    e = hipMemcpy(back, slots, sizeof(back), hipMemcpyDeviceToHost);
    if (absmax_out_host) *absmax_out_host = back[1];
    if (range_flag_out_host) std::memcpy(range_flag_out_host, &back[2], 4);
  }
  RAMP_HIP_CHECK(e);
  return rc;
}

int ramp_op_tkl(const float* X, const float* W, const float* bias, const float* resid, const float* rowbias,
                const int32_t* rowvar, int32_t n_var, int32_t L, const float* ln_g, const float* ln_b, int32_t M, int32_t N,
                float absmax_prev, float* Y, float* absmax_out_host, int32_t* range_flag_out_host, void* stream) {
  return op_tkl_impl(X, W, bias, resid, rowbias, rowvar, n_var, L, ln_g, ln_b, M, N, absmax_prev, Y, absmax_out_host, range_flag_out_host, stream, false);
}
int ramp_op_tkl16(const float* X, const float* W, const float* bias, const float* resid, const float* rowbias,
                  const int32_t* rowvar, int32_t n_var, int32_t L, const float* ln_g, const float* ln_b, int32_t M, int32_t N,
                  float absmax_prev, float* Y, float* absmax_out_host, int32_t* range_flag_out_host, void* stream) {
  return op_tkl_impl(X, W, bias, resid, rowbias, rowvar, n_var, L, ln_g, ln_b, M, N, absmax_prev, Y, absmax_out_host, range_flag_out_host, stream, true);
}

int ramp_op_ato(const float* qkv, const float* Wo, const float* bias, const float* resid, const float* rowbias, const int32_t* rowvar,
                int32_t n_var, int32_t L, int32_t M, float absmax_prev, float* Y, float* absmax_out_host, int32_t* range_flag_out_host, void* stream) {
  RAMP_REQUIRE(qkv && Wo && resid && Y && M > 0 && L > 0, "bad arguments");
  hipStream_t s = as_stream(stream);
  DevArena ar;
  std::vector<float> hw((size_t)256 * 256);
  RAMP_HIP_CHECK(hipMemcpy(hw.data(), Wo, hw.size() * 4, hipMemcpyDeviceToHost));
  float mx = 0.f;
  for (float v : hw) mx = std::max(mx, std::fabs(v));
  float sc = 1.f;
  if (mx > 0.f && std::isfinite(mx)) { int e; std::frexp(mx, &e); sc = std::ldexp(1.f, 11 - e); }
  unsigned short* stream_w = reinterpret_cast<unsigned short*>(ar.alloc(8 * 8192 + 4));
  float* slots = ar.alloc(4);
  RAMP_REQUIRE(stream_w && slots, "hipMalloc failed");
  CK(init_atk_attributes());
  CK(ato_pack(Wo, sc, stream_w, s));
  const float host[4] = {absmax_prev, 0.f, 0.f, 0.f};
  RAMP_HIP_CHECK(hipMemcpyAsync(slots, host, sizeof(host), hipMemcpyHostToDevice, s));
  AtoArgs a; a.M = M; a.L = L; a.QKV = qkv; a.W = stream_w; a.bias = bias; a.resid = resid; a.Y = Y;
  a.rowbias = rowbias; a.rowvar = rowvar; a.row0 = 0; a.rb_stride = 256; a.n_var = rowbias ? n_var : 0;
  a.amax_in = absmax_prev > 0.f ? slots : nullptr; a.amax_out = slots + 1; a.wsi = 1.f / sc; a.site = 0;
  a.range_flag = reinterpret_cast<int*>(slots + 2);
  int rc = launch_ato(a, s);
  hipError_t e = hipStreamSynchronize(s);
  float back[4] = {0, 0, 0, 0};
  if (rc == 0 && e == hipSuccess) {
    e = hipMemcpy(back, slots, sizeof(back), hipMemcpyDeviceToHost);
    if (absmax_out_host) *absmax_out_host = back[1];
    if (range_flag_out_host) std::memcpy(range_flag_out_host, &back[2], 4);
  }
  RAMP_HIP_CHECK(e);
  return rc;
}

int ramp_op_atb(const float* qkv, const float* dout, float* dqkv, int32_t M, int32_t L, void* stream) {
  RAMP_REQUIRE(qkv && dout && dqkv && M > 0 && L > 0, "bad arguments");
  AtbArgs a; a.M = M; a.L = L; a.QKV = qkv; a.dO = dout; a.dQKV = dqkv;
  return launch_atb(a, as_stream(stream));
}

int ramp_op_abl(const float* qkv, const float* dout, const float* W, const float* z, const float* ln_g, const float* add, int32_t M, int32_t L,
                float absmax_prev, float* out, float* absmax_out_host, int32_t* range_flag_out_host, void* stream) {
  RAMP_REQUIRE(qkv && dout && W && z && ln_g && add && out && M > 0 && L > 0, "bad arguments");
  hipStream_t s = as_stream(stream);
  DevArena ar;
  std::vector<float> hw((size_t)256 * 768);
  RAMP_HIP_CHECK(hipMemcpy(hw.data(), W, hw.size() * 4, hipMemcpyDeviceToHost));
  float mx = 0.f;
  for (float v : hw) mx = std::max(mx, std::fabs(v));
  float sc = 1.f;
  if (mx > 0.f && std::isfinite(mx)) { int e; std::frexp(mx, &e); sc = std::ldexp(1.f, 11 - e); }
  unsigned short* stream_w = reinterpret_cast<unsigned short*>(ar.alloc((size_t)256 * 768 + 4));
  float* slots = ar.alloc(4);
  RAMP_REQUIRE(stream_w && slots, "hipMalloc failed");
  CK(init_atl_attributes());
  CK(abl_pack(W, sc, stream_w, s));
  const float host[4] = {absmax_prev, 0.f, 0.f, 0.f};
  RAMP_HIP_CHECK(hipMemcpyAsync(slots, host, sizeof(host), hipMemcpyHostToDevice, s));
  AblArgs a; a.M = M; a.L = L; a.QKV = qkv; a.dO = dout; a.W = stream_w; a.Z = z; a.add = add; a.ln_g = ln_g; a.Y = out;
  a.amax_in = absmax_prev > 0.f ? slots : nullptr; a.amax_out = slots + 1; a.wsi = 1.f / sc; a.site = 0;
  a.range_flag = reinterpret_cast<int*>(slots + 2);
  int rc = launch_abl(a, s);
  hipError_t e = hipStreamSynchronize(s);
  float back[4] = {0, 0, 0, 0};
  if (rc == 0 && e == hipSuccess) {
    e = hipMemcpy(back, slots, sizeof(back), hipMemcpyDeviceToHost);
    if (absmax_out_host) *absmax_out_host = back[1];
    if (range_flag_out_host) std::memcpy(range_flag_out_host, &back[2], 4);
  }
  RAMP_HIP_CHECK(e);
  return rc;
}

int ramp_op_tkw(const float* X, const float* X2, int32_t K1, const float* W, const float* bias, const float* resid, const float* resid2,
                const float* gn_c, const float* gn_stats, const float* gn_gamma, const float* gn_beta, const float* gamma, const float* beta,
                const float* tbias, int32_t M, int32_t L, int32_t N, int32_t K, int32_t dir, int32_t N1, float absmax_prev, float* Y, float* Y2,
                float* Cst, float* stats, float* absmax_out_host, int32_t* range_flag_out_host, void* stream) {
  RAMP_REQUIRE(X && W && Y && M > 0 && L > 0 && N % 32 == 0 && K % 16 == 0, "bad arguments");
  hipStream_t s = as_stream(stream);
  DevArena ar;
  const size_t n = (size_t)5 * N * K;
  if (N <= 64 && K <= 64) {      // the narrow layers: the same fusion on sample-owning WAVES (tkc.hip)
    RAMP_REQUIRE(!X2 && !Y2 && tkc_applicable(M, L, N, K, nullptr), "narrow fused convolution: C_in, C_out in {32, 64}, L >= 8 dividing 48 or 32, one operand, one output");
    std::vector<float> hw(n);
    RAMP_HIP_CHECK(hipMemcpy(hw.data(), W, n * 4, hipMemcpyDeviceToHost));
    float mx = 0.f; for (float v : hw) mx = std::max(mx, std::fabs(v));
    float sc = 1.f;
    if (mx > 0.f && std::isfinite(mx)) { int e; std::frexp(mx, &e); sc = std::ldexp(1.f, 11 - e); }
    unsigned short* pl = reinterpret_cast<unsigned short*>(ar.alloc(tkc_packed_halves(N, K) / 2 + 4));
    float* sl = ar.alloc(4);
    RAMP_REQUIRE(pl && sl, "hipMalloc failed");
    CK(init_tkc_attributes());
    CK(tkc_pack(W, N, K, sc, pl, s));
    const float v[4] = {absmax_prev, 0.f, 0.f, 0.f};
    RAMP_HIP_CHECK(hipMemcpyAsync(sl, v, 16, hipMemcpyHostToDevice, s));
    TkcArgs t; t.M = M; t.L = L; t.N = N; t.K = K; t.dir = dir; t.X = X; t.ldx = K; t.W = pl; t.bias = bias; t.resid = resid; t.ldr = N;
    t.resid2 = resid2; t.ldr2 = N; t.Y = Y; t.ldy = N; t.amax_in = absmax_prev > 0.f ? sl : nullptr; t.amax_out = sl + 1; t.wsi = 1.f / sc;
    t.range_flag = reinterpret_cast<int*>(sl + 2);
    t.gn_c = gn_c; t.gn_stats = gn_stats; t.gn_gamma = gn_gamma; t.gn_beta = gn_beta;
    t.Cst = Cst; t.stats = stats; t.gamma = gamma; t.beta = beta; t.tbias = tbias; t.eps = 1e-5f;
    int rc5 = launch_tkc(t, s);
    hipError_t e5 = hipStreamSynchronize(s);
    float back[4] = {0, 0, 0, 0};
    if (rc5 == 0 && e5 == hipSuccess) {
      e5 = hipMemcpy(back, sl, sizeof(back), hipMemcpyDeviceToHost);
      if (absmax_out_host) *absmax_out_host = back[1];
      if (range_flag_out_host) std::memcpy(range_flag_out_host, &back[2], 4);
    }
    RAMP_HIP_CHECK(e5);
    return rc5;
  }
  std::vector<float> hw(n);
  RAMP_HIP_CHECK(hipMemcpy(hw.data(), W, n * 4, hipMemcpyDeviceToHost));
  float mx = 0.f;
  for (float v : hw) mx = std::max(mx, std::fabs(v));
  float sc = 1.f;
  if (mx > 0.f && std::isfinite(mx)) { int e; std::frexp(mx, &e); sc = std::ldexp(1.f, 11 - e); }
  unsigned short* planes = reinterpret_cast<unsigned short*>(ar.alloc(n + 4));
  float* slots = ar.alloc(4);
  RAMP_REQUIRE(planes && slots, "hipMalloc failed");
  CK(init_tkw_attributes());
  CK(launch_pack_h3(W, planes, (long)5 * N, K, sc, s));
  const float host[4] = {absmax_prev, 0.f, 0.f, 0.f};
  RAMP_HIP_CHECK(hipMemcpyAsync(slots, host, sizeof(host), hipMemcpyHostToDevice, s));
  TkwArgs a; a.M = M; a.L = L; a.N = N; a.K = K; a.dir = dir; a.X = X; a.ldx = X2 ? K1 : K; a.X2 = X2; a.ldx2 = X2 ? K - K1 : 0; a.K1 = X2 ? K1 : K;
  a.gn_c = gn_c; a.gn_stats = gn_stats; a.gn_gamma = gn_gamma; a.gn_beta = gn_beta; a.W = planes; a.wsi = 1.f / sc; a.bias = bias;
  a.resid = resid; a.ldr = N; a.resid2 = resid2; a.ldr2 = N; a.Y = Y; a.ldy = Y2 ? N1 : N; a.Y2 = Y2; a.ldy2 = Y2 ? N - N1 : 0; a.N1 = Y2 ? N1 : N;
  a.Cst = Cst; a.stats = stats; a.gamma = gamma; a.beta = beta; a.tbias = tbias; a.eps = 1e-5f;
  a.amax_in = absmax_prev > 0.f ? slots : nullptr; a.amax_out = slots + 1; a.site = 0; a.range_flag = reinterpret_cast<int*>(slots + 2);
  int rc = launch_tkw(a, s);
  hipError_t e = hipStreamSynchronize(s);
  float back[4] = {0, 0, 0, 0};
  if (rc == 0 && e == hipSuccess) {
    e = hipMemcpy(back, slots, sizeof(back), hipMemcpyDeviceToHost);
    if (absmax_out_host) *absmax_out_host = back[1];
    if (range_flag_out_host) std::memcpy(range_flag_out_host, &back[2], 4);
  }
  RAMP_HIP_CHECK(e);
  return rc;
}

int ramp_op_tklb(const float* dqkv, const float* W, const float* z, const float* ln_g, const float* add, int32_t M,
                 float absmax_prev, float* out, float* absmax_out_host, int32_t* range_flag_out_host, void* stream) {
  RAMP_REQUIRE(dqkv && W && z && ln_g && add && out && M > 0, "bad arguments");
  hipStream_t s = as_stream(stream);
  DevArena ar;
  std::vector<float> hw((size_t)256 * 768);
  RAMP_HIP_CHECK(hipMemcpy(hw.data(), W, hw.size() * 4, hipMemcpyDeviceToHost));
  float mx = 0.f;
  for (float v : hw) mx = std::max(mx, std::fabs(v));
  float sc = 1.f;
  if (mx > 0.f && std::isfinite(mx)) { int e; std::frexp(mx, &e); sc = std::ldexp(1.f, 11 - e); }
  unsigned short* planes = reinterpret_cast<unsigned short*>(ar.alloc((size_t)256 * 768 + 4));
  float* slots = ar.alloc(4);
  RAMP_REQUIRE(planes && slots, "hipMalloc failed");
  CK(launch_pack_h3(W, planes, 256, 768, sc, s));
  const float host[4] = {absmax_prev, 0.f, 0.f, 0.f};
  RAMP_HIP_CHECK(hipMemcpyAsync(slots, host, sizeof(host), hipMemcpyHostToDevice, s));
  TklbArgs a; a.M = M; a.X = dqkv; a.Z = z; a.add = add; a.Y = out; a.W = planes; a.ln_g = ln_g;
  a.amax_in = absmax_prev > 0.f ? slots : nullptr; a.amax_out = slots + 1; a.wsi = 1.f / sc; a.site = 0;
  a.range_flag = reinterpret_cast<int*>(slots + 2);
  int rc = launch_tklb(a, s);
  hipError_t e = hipStreamSynchronize(s);
  float back[4] = {0, 0, 0, 0};
  if (rc == 0 && e == hipSuccess) {
    e = hipMemcpy(back, slots, sizeof(back), hipMemcpyDeviceToHost);
    if (absmax_out_host) *absmax_out_host = back[1];
    if (range_flag_out_host) std::memcpy(range_flag_out_host, &back[2], 4);
  }
  RAMP_HIP_CHECK(e);
  return rc;
}

int ramp_op_groupnorm(const float* x, const float* gamma, const float* beta, const float* tbias, const float* resid,
                      float* y, float* stats, int32_t R, int32_t L, int32_t C, float eps, int32_t mish, void* stream) {
  RAMP_REQUIRE(x && gamma && beta && y, "null argument");
  GnArgs g; g.x = x; g.gamma = gamma; g.beta = beta; g.tbias = tbias; g.resid = resid; g.y = y; g.stats = stats;
  g.R = R; g.L = L; g.C = C; g.eps = eps; g.mish = mish;
  return launch_gn_fwd(g, as_stream(stream));
}
int ramp_op_groupnorm_bwd(const float* dy, const float* x, const float* stats, const float* gamma, const float* beta,
                          const float* add, float* dx, int32_t R, int32_t L, int32_t C, int32_t mish, void* stream) {
  RAMP_REQUIRE(dy && x && stats && gamma && beta && dx, "null argument");
  GnBwdArgs g; g.dy = dy; g.x = x; g.stats = stats; g.gamma = gamma; g.beta = beta; g.add = add; g.dx = dx;
  g.R = R; g.L = L; g.C = C; g.mish = mish;
  return launch_gn_bwd(g, as_stream(stream));
}
int ramp_op_layernorm(const float* x, const float* gamma, const float* beta, float* y, int32_t n_tok, void* stream) {
  RAMP_REQUIRE(x && gamma && beta && y, "null argument");
  return launch_ln_fwd(x, gamma, beta, y, n_tok, as_stream(stream));
}
int ramp_op_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* add, float* dx,
                          int32_t n_tok, void* stream) {
  RAMP_REQUIRE(dy && x && gamma && dx, "null argument");
  return launch_ln_bwd(dy, x, gamma, add, dx, n_tok, as_stream(stream));
}
int ramp_op_geglu(const float* ag, float* hg, int32_t n_tok, int32_t F, void* stream) {
  RAMP_REQUIRE(ag && hg, "null argument");
  return launch_geglu_fwd(ag, hg, n_tok, F, as_stream(stream));
}
int ramp_op_geglu_bwd(const float* dhg, const float* ag, float* dag, int32_t n_tok, int32_t F, void* stream) {
  RAMP_REQUIRE(dhg && ag && dag, "null argument");
  return launch_geglu_bwd(dhg, ag, dag, n_tok, F, as_stream(stream));
}
int ramp_op_attention(const float* qkv, float* o, int32_t R, int32_t L, void* stream) {
  RAMP_REQUIRE(qkv && o, "null argument");
  return launch_attn_fwd(qkv, o, R, L, as_stream(stream));
}
int ramp_op_attention_bwd(const float* qkv, const float* dout, float* dqkv, int32_t R, int32_t L, void* stream) {
  RAMP_REQUIRE(qkv && dout && dqkv, "null argument");
  return launch_attn_bwd(qkv, dout, dqkv, R, L, as_stream(stream));
}


}  // extern "C"

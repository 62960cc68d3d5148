// Context, weight packing and the forward / input-VJP schedule of the RAMP score network,
// plus the sampler loops and their hipGraph capture.  Host-side C++ over the kernels in
// gemm.hip / rowops.hip / sampler.hip; exported through the C ABI in include/ramp_hip.h.
//
// Reference structure being scheduled: TemporalUnetInference.forward_no_energy
// (UnetInference.py:176-224) and EnergyGradFunction (UnetInference.py:19-37) whose
// autograd.grad is replaced by the explicit dX chain below (all parameters are frozen, so no
// dW / db products exist anywhere).
#include "engine_util.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace ramp {

static thread_local std::string g_err;
void set_last_error(const std::string& msg) { g_err = msg; }
int device_cu_count() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (!cached[dev]) {
    hipDeviceProp_t p;
    cached[dev] = hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  }
  return cached[dev];
}
const char* last_error_cstr() { return g_err.c_str(); }

struct ConvW {       // k=5 conv as 5-tap GEMM
  float* fwd = nullptr;   // [5][Cout][Cin]
  float* bwd = nullptr;   // [5][Cin][Cout]
  float* bias = nullptr;
  int cin = 0, cout = 0;
};
struct RTB {
  std::string name;
  int cin = 0, cout = 0, L = 0, tb_off = 0;
  bool has_res = false, first = false;
  ConvW c1, c2;
  float *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr;
  float *res_f = nullptr, *res_b = nullptr, *res_bias = nullptr;   // [Cout][Cin], [Cin][Cout]
  float *w5in = nullptr, *w1in = nullptr;                           // first layer packed [5][S][32], [S][32]
  // activations (capacity rows)
  float *a_c1 = nullptr, *a_st1 = nullptr, *a_h = nullptr, *a_c2 = nullptr, *a_st2 = nullptr, *a_out = nullptr;
};
struct STBlock {
  float *wqkv_f = nullptr, *wqkv_b = nullptr, *wo_f = nullptr, *wo_b = nullptr, *bo = nullptr;
  float *w1_f = nullptr, *w1_b = nullptr, *b1 = nullptr, *w2_f = nullptr, *w2_b = nullptr, *b2 = nullptr;
  float *w1_pk = nullptr, *b1_pk = nullptr;        // GEGLU-packed forward weight / bias
  float *ln1_g = nullptr, *ln1_b = nullptr, *ln3_g = nullptr, *ln3_b = nullptr;
  const float *wv2 = nullptr, *wo2 = nullptr, *bo2 = nullptr;    // attn2 (cross) raw
  float *a_qkv = nullptr, *a_z1 = nullptr, *a_ag = nullptr, *a_z2 = nullptr;
  // token-owning fused feed-forward (ffx.hip): the two weight streams, the weight scales of their first / second product
  unsigned short *ffx_f = nullptr, *ffx_b = nullptr; float ffx_wsi_w1 = 1.f, ffx_wsi_w2 = 1.f;
  unsigned short *ffx16_f = nullptr, *ffx16_b = nullptr;     // the same streams in 16 x 32 fragments (ffx16.hip)
  // self-attention fused with the output projection (atk.hip): the projection's weight stream
  unsigned short* ato_w = nullptr; float ato_wsi = 1.f;
  unsigned short* abl_w = nullptr;       // weight stream of the fused attention backward + d(ln1) (atl.hip), scale = the wqkv_b planes'
};
struct ST {
  std::string name;
  int C = 0, L = 0, blk0 = 0;
  float *gn_g = nullptr, *gn_b = nullptr, *wpi_f = nullptr, *wpi_b = nullptr, *bpi = nullptr;
  float *wpo_f = nullptr, *wpo_b = nullptr, *bpo = nullptr;
  STBlock blk[2];
  float *a_gst = nullptr, *a_z0 = nullptr, *a_y = nullptr;
};
struct Resample {
  std::string name;
  float *w_f = nullptr, *w_b = nullptr, *bias = nullptr;
  int C = 0, Lin = 0, Lout = 0, taps = 0;
  float* a_y = nullptr;
};

}  // namespace ramp

using namespace ramp;

#ifndef RAMP_DEFAULT_GEMM_MODE
#define RAMP_DEFAULT_GEMM_MODE 0
#endif

struct ramp_ctx {
  ramp_config cfg{};
  int device = 0;
  DevArena arena;
  std::unordered_map<std::string, std::pair<float*, std::vector<int64_t>>> raw;   // device copies
  bool finalized = false;
  std::vector<int> chan;       // channels per level
  std::vector<RTB> rtbs;       // order: downs (2/level), mid1, mid2, ups (2/level)
  std::vector<ST> sts;         // order: downs, mid, ups
  std::vector<Resample> downs, ups;
  ConvW final_conv; float *fin_g = nullptr, *fin_b = nullptr, *fin_w = nullptr, *fin_bias = nullptr;
  float *a_fin_c = nullptr, *a_fin_st = nullptr, *a_fin_a = nullptr, *a_fin_da = nullptr;
  // shared temporaries
  float *t_res = nullptr, *t_xn = nullptr, *t_ln = nullptr, *t_o = nullptr, *t_hg = nullptr;
  float *t_dag = nullptr, *t_dqkv = nullptr, *t_dln = nullptr, *t_dz = nullptr, *t_dz1 = nullptr;
  float *g_a = nullptr, *g_b = nullptr, *g_t1 = nullptr, *g_t2 = nullptr, *g_tr = nullptr;
  std::vector<float*> skip_grad;   // per level
  // time table / scene
  float* time_table = nullptr; int tt_stride = 0, tt_T = 0; float* time_temb = nullptr;
  float* cross_bias = nullptr; int n_variants = 0; int* row_variant = nullptr; int row_variant_cap = 0;
  int n_blocks_total = 0;
  // device pointer tables for setup kernels
  void* d_ptr_tables = nullptr;
  // sampler state
  float *s_x = nullptr, *s_eps = nullptr, *s_mean = nullptr, *s_x0 = nullptr, *s_noise = nullptr, *s_chain = nullptr;
  size_t s_cap_B = 0, s_cap_rows = 0, s_noise_cap = 0, s_chain_cap = 0;
  unsigned long long* s_philox = nullptr;      // device {seed, offset} of a job that draws its own noise (ramp_sample_params.noise_mode 1)
  int* s_hard_idx = nullptr; float* s_hard_val = nullptr; size_t s_hard_val_cap = 0; float* s_window = nullptr;
  float* s_cloud = nullptr; size_t s_cloud_cap = 0;
  // graph cache
  hipGraphExec_t graph_exec[2] = {nullptr, nullptr}; std::string graph_key;   // [0] first evaluation calibrates, [1] it continues
  // Round 6: a job the range guard flagged is repeated IN fp16x3 (ramp_set_fallback(ctx, 2)): the guard's state after every evaluation of a
  // job is logged on the device (trip_log[j]), ramp_range_status finds the first flagged evaluation, and the repeat runs exactly that
  // evaluation as a calibrating one (bf16x6 kernels that record every call site's operand maximum: range-free, and its successor is scaled
  // from maxima that are true) -- a function of the job alone, 1.03 x a steady job instead of the 1.83 x of an all-bf16x6 repeat
  int* trip_log = nullptr; int trip_log_cap = 0; int last_job_steps = 0; int trip_eval = -1; int trip_site = -1; int rerun = 0;
  hipGraphExec_t graph_rerun = nullptr; std::string graph_rerun_key;
  // fp16x3 calibration kept from one ramp_sample to the next of the same job shape (ramp_set_calibration_reuse): the
  // job's first evaluation then reads the maxima the previous job's first evaluation recorded (table 2).  A job becomes
  // the next one's calibration only when ramp_range_status has reported it clean.
  int cal_reuse = 1; bool s_calibrated = false, s_pending = false; std::string s_cal_key, s_pending_key;
  // Round 5: the calibration a job's first evaluation is scaled from is CANONICAL -- one bf16x6 evaluation, outside the job, on Philox noise of
  // a fixed seed with the job's hard conditions, at the job's first timestep (table 2) -- instead of whatever the previous job left behind: no
  // job contains a bf16x6 evaluation, and a job's result does not depend on what ran before it on the context (ramp_set_calibration_reuse)
  bool c_cal_valid = false; std::string c_cal_key; unsigned long long* c_cal_rec = nullptr;
  std::map<std::string, float*> c_cal_saved;     // canonical tables already computed, by key: switching between job shapes copies 4 KB instead of re-evaluating
  std::vector<float*> c_cal_free;                // their buffers after an invalidation (scene / plan change), for reuse
  // receding-horizon replanning (ramp_replan): fixed device buffers the captured graphs read, the graph of a replan whose
  // first evaluation calibrates ([0]) and of one that continues from the previous replan's operand maxima ([1])
  float *r_noise = nullptr, *r_hist = nullptr, *r_xclean = nullptr, *r_best = nullptr, *r_plen = nullptr, *r_smooth = nullptr,
        *r_cost = nullptr, *r_hard_val = nullptr;
  double *r_static = nullptr, *r_dyn = nullptr;
  int *r_mask = nullptr, *r_en = nullptr, *r_result = nullptr, *r_hard_idx = nullptr;
  ReplanState* r_state = nullptr;
  size_t r_cap_B = 0, r_cap_static = 0, r_cap_dyn = 0, r_cap_cost = 0;
  hipGraphExec_t r_graph[2] = {nullptr, nullptr}; std::string r_key;
  bool r_calibrated = false;
  // scene-encoder scratch
  float* scene_ws = nullptr; size_t scene_ws_cap = 0;
  // bf16x6 weight planes: fp32 weight base pointer -> (planes, element count)
  int gemm_mode = 0;                 // 0 = exact fp32 MFMA, 1 = bf16x6 split on the bf16 matrix cores
  struct X6W { unsigned short* planes; size_t n; int K; unsigned short* packed; unsigned short* packed3; float w_scale_inv; };
  std::map<const float*, X6W> x6;
  int x6_pipe = 1;                   // 1 = fragment-packed weights + pipelined kernel (RAMP_X6_PIPE=0: LDS-staged weights)
  int geglu_group = 64;              // GEGLU weight tiling (32 with the pipelined kernels' wave-private epilogue)
  // fp16x3 (gemm_mode 2): delayed operand scaling.  phase 0 = bf16x6; 1 = bf16x6 that records max|A| per GEMM call site
  // (the calibration evaluation: the first score evaluation of every ramp_sample); 2 = fp16x3 scaled from the
  // previous evaluation's maxima, recording its own.  obs[2][MAX_SITES] floats, ping-pong by evaluation.
  static constexpr int MAX_SITES = 1024, N_OBS_TABLES = 4;      // 0 / 1: ping-pong by evaluation, 2: the canonical calibration of the sampling jobs, 3: carried from replan to replan
  int phase = 0, site = 0;
  int ff_fused = 150000;             // fp16x3 evaluations: FF1 -> GEGLU -> FF2 as one launch for M >= this many rows
                                     // (RAMP_FF_FUSED: 0 never, 1 always, n > 1 that threshold)
  int three_blocks = 1;              // launch plan: third resident block for the bias-only linears
  int mfma16 = 1;                    // launch plan: the token-owning fused feed-forward on v_mfma_f32_16x16x32_f16 (ffx16.hip; 0: the 32x32x16 kernels of
                                     // ffx.hip) -- the shape that holds the higher clock under the power cap (RAMP_MFMA16)
  int ffx_min_rows = 32768;          // fp16x3 evaluations: feed-forward pairs with at least this many tokens run the token-owning fused
                                     // kernels of ffx.hip, forward and backward (RAMP_FFX: 0 never, n that threshold)
  int tkl_min_rows = 65536;          // fp16x3 evaluations: K = 256 transformer linears (LN1 -> QKV, out-proj, d(o)) with at least this many
                                     // tokens run the token-owning kernel of tkl.hip (RAMP_TKL: 0 never, n that threshold)
  int atk_min_rows = 40000;          // fp16x3 evaluations: self-attention + output projection as one launch of sample-owning waves (atk.hip)
                                     // from this many tokens where the level's token count divides 48 or 32 (RAMP_ATK: 0 never, n that threshold)
  int tkc_gn = 3;                    // ... and the GroupNorm + Mish of their Conv1dBlock inside the same launch: bit 0 forward (epilogue), bit 1 input
                                     // gradient (operand); RAMP_TKC_GN=0: separate gn kernels
  int tkc_min_rows = 16384;          // fp16x3 evaluations: the k = 5 convolutions with C_in, C_out in {32, 64} as sample-owning waves (tkc.hip) from this
                                     // many tokens, on levels whose token count (>= 8) divides 48 or 32 (RAMP_TKC: 0 never, n that threshold)
  int tkw_min_rows = 16384;          // fp16x3 evaluations: the k = 5 convolutions with C_out in {128, 256, 512} with GroupNorm + Mish fused around them as
                                     // sample-owning blocks (tkw.hip) from this many tokens, on levels whose token count (>= 3) divides 96 (RAMP_TKW)
  std::map<const float*, unsigned short*> tk16_w;          // K = 256 linear weight (fp32 base pointer) -> its 16 x 32 fragment planes (tkl16.hip; scale = the tile kernels' planes')
  struct TkcW { unsigned short* planes; float wsi; };
  std::map<const float*, TkcW> tkc_w;                       // fp32 conv weight [5][N][K] -> its tkc planes
  int share_prefix = 1;              // sampling jobs: rows of one trajectory share the network prefix (RAMP_SHARE_PREFIX=0: off)
  int force_x6 = 0;                  // ramp_set_fallback: run ramp_sample entirely in bf16x6 although the mode is fp16x3
  // single evaluations (ramp_score): the tables of the last evaluation stay valid as the next one's calibration
  bool score_calibrated = false; bool score_calibrated_bwd = false; int score_parity = 0; int score_last_mode = 0;
  float* obs = nullptr; float *obs_in = nullptr, *obs_out = nullptr;
  int* range_flag = nullptr;
  // debug
  std::map<std::string, std::pair<float*, size_t>> dbg;
  int64_t launches = 0;
  // per-launch HIP-event profiler (eager mode only): category, algorithmic flops, start/stop events
  bool prof_on = false, prof_dump = false; int ffx_ablate = 0; bool tklb_off = false; bool abl_on = true;
  std::vector<hipEvent_t> prof_ev; size_t prof_used = 0;
  std::vector<int> prof_cat; std::vector<double> prof_flops; std::vector<std::array<int, 4>> prof_shape;
};

enum { CAT_GEMM = 0, CAT_ATTN = 1, CAT_ROW = 2, CAT_SMALLCONV = 3, CAT_SAMPLER = 4, CAT_N = 5 };

static void prof_pre(ramp_ctx* c, hipStream_t s, int cat, double flops, std::array<int, 4> shape = {0, 0, 0, 0}) {
  if (!c->prof_on) return;
  c->prof_shape.push_back(shape);
  if (c->prof_used + 2 > c->prof_ev.size()) {
    const size_t old = c->prof_ev.size();
    c->prof_ev.resize(old + 4096);
    for (size_t i = old; i < c->prof_ev.size(); ++i) (void)hipEventCreate(&c->prof_ev[i]);
  }
  c->prof_cat.push_back(cat); c->prof_flops.push_back(flops);
  (void)hipEventRecord(c->prof_ev[c->prof_used++], s);
}
static void prof_post(ramp_ctx* c, hipStream_t s) {
  if (!c->prof_on) return;
  (void)hipEventRecord(c->prof_ev[c->prof_used++], s);
}
// run one kernel launch expression with launch counting and optional event bracketing
#define LAUNCH(ctx_, stream_, cat_, flops_, expr_)                     \
  do {                                                                 \
    prof_pre((ctx_), (stream_), (cat_), (double)(flops_));             \
    int _rc = (expr_);                                                 \
    prof_post((ctx_), (stream_));                                      \
    (ctx_)->launches++;                                                \
    if (_rc != 0) return _rc;                                          \
  } while (0)

namespace {


int dev_alloc(ramp_ctx* c, float** out, size_t n) {
  *out = c->arena.alloc(n);
  RAMP_REQUIRE(*out != nullptr, "hipMalloc failed for " + std::to_string(n * 4) + " bytes");
  return 0;
}

int get_raw(ramp_ctx* c, const std::string& key, std::initializer_list<int64_t> shape, float** out) {
  auto it = c->raw.find(key);
  RAMP_REQUIRE(it != c->raw.end(), "missing weight '" + key + "'");
  std::vector<int64_t> want(shape);
  RAMP_REQUIRE(it->second.second == want, "weight '" + key + "' has the wrong shape");
  *out = it->second.first;
  return 0;
}

int permute3(ramp_ctx* c, const float* in, int D0, int D1, int D2, int p0, int p1, int p2, float** out) {
  CK(dev_alloc(c, out, (size_t)D0 * D1 * D2));
  hipLaunchKernelGGL(permute3_kernel, dim3(std::min<long>(((long)D0 * D1 * D2 + 255) / 256, 4096)), dim3(256), 0, 0,
                     in, *out, D0, D1, D2, p0, p1, p2);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

int pack_conv5(ramp_ctx* c, const std::string& prefix, int cin, int cout, ConvW* w) {
  float* raw; float* b;
  CK(get_raw(c, prefix + ".weight", {cout, cin, 5}, &raw));
  CK(get_raw(c, prefix + ".bias", {cout}, &b));
  w->cin = cin; w->cout = cout; w->bias = b;
  CK(permute3(c, raw, cout, cin, 5, 2, 0, 1, &w->fwd));   // [5][Cout][Cin]
  CK(permute3(c, raw, cout, cin, 5, 2, 1, 0, &w->bwd));   // [5][Cin][Cout]
  return 0;
}

int pack_linear(ramp_ctx* c, const std::string& key, int n, int k, bool conv1, float** fwd, float** bwd) {
  float* raw;
  if (conv1) CK(get_raw(c, key, {n, k, 1}, &raw)); else CK(get_raw(c, key, {n, k}, &raw));
  *fwd = raw;
  CK(permute3(c, raw, n, k, 1, 1, 0, 2, bwd));            // [K][N]
  return 0;
}

int build_rtb(ramp_ctx* c, RTB& r) {
  const std::string& n = r.name;
  if (r.first) {
    float* raw;
    CK(get_raw(c, n + ".blocks.0.block.0.weight", {r.cout, r.cin, 5}, &raw));
    CK(get_raw(c, n + ".blocks.0.block.0.bias", {r.cout}, &r.c1.bias));
    CK(permute3(c, raw, r.cout, r.cin, 5, 2, 1, 0, &r.w5in));            // [5][S][32]
    CK(get_raw(c, n + ".residual_conv.weight", {r.cout, r.cin, 1}, &raw));
    CK(permute3(c, raw, r.cout, r.cin, 1, 2, 1, 0, &r.w1in));            // [1][S][32]
    CK(get_raw(c, n + ".residual_conv.bias", {r.cout}, &r.res_bias));
    r.c1.cin = r.cin; r.c1.cout = r.cout;
  } else {
    CK(pack_conv5(c, n + ".blocks.0.block.0", r.cin, r.cout, &r.c1));
    if (r.has_res) {
      CK(pack_linear(c, n + ".residual_conv.weight", r.cout, r.cin, true, &r.res_f, &r.res_b));
      CK(get_raw(c, n + ".residual_conv.bias", {r.cout}, &r.res_bias));
    }
  }
  CK(pack_conv5(c, n + ".blocks.1.block.0", r.cout, r.cout, &r.c2));
  CK(get_raw(c, n + ".blocks.0.block.2.weight", {r.cout}, &r.g1));
  CK(get_raw(c, n + ".blocks.0.block.2.bias", {r.cout}, &r.b1));
  CK(get_raw(c, n + ".blocks.1.block.2.weight", {r.cout}, &r.g2));
  CK(get_raw(c, n + ".blocks.1.block.2.bias", {r.cout}, &r.b2));
  const size_t cap = (size_t)c->cfg.max_rows, lc = (size_t)r.L * r.cout;
  CK(dev_alloc(c, &r.a_c1, cap * lc)); CK(dev_alloc(c, &r.a_st1, cap * 16));
  CK(dev_alloc(c, &r.a_h, cap * lc));  CK(dev_alloc(c, &r.a_c2, cap * lc));
  CK(dev_alloc(c, &r.a_st2, cap * 16)); CK(dev_alloc(c, &r.a_out, cap * lc));
  return 0;
}

int build_st(ramp_ctx* c, ST& s) {
  const std::string& n = s.name;
  const int D = 256, ctx = c->cfg.context_dim;
  CK(get_raw(c, n + ".norm.weight", {s.C}, &s.gn_g));
  CK(get_raw(c, n + ".norm.bias", {s.C}, &s.gn_b));
  CK(pack_linear(c, n + ".proj_in.weight", D, s.C, true, &s.wpi_f, &s.wpi_b));
  CK(get_raw(c, n + ".proj_in.bias", {D}, &s.bpi));
  CK(pack_linear(c, n + ".proj_out.weight", s.C, D, true, &s.wpo_f, &s.wpo_b));
  CK(get_raw(c, n + ".proj_out.bias", {s.C}, &s.bpo));
  const size_t cap = (size_t)c->cfg.max_rows, tok = cap * s.L;
  CK(dev_alloc(c, &s.a_gst, cap * 16)); CK(dev_alloc(c, &s.a_z0, tok * D)); CK(dev_alloc(c, &s.a_y, cap * s.L * s.C));
  for (int b = 0; b < 2; ++b) {
    STBlock& k = s.blk[b];
    const std::string t = n + ".transformer_blocks." + std::to_string(b);
    float *q, *kk, *v;
    CK(get_raw(c, t + ".attn1.to_q.weight", {D, D}, &q));
    CK(get_raw(c, t + ".attn1.to_k.weight", {D, D}, &kk));
    CK(get_raw(c, t + ".attn1.to_v.weight", {D, D}, &v));
    CK(dev_alloc(c, &k.wqkv_f, 3 * D * D));
    RAMP_HIP_CHECK(hipMemcpy(k.wqkv_f, q, D * D * 4, hipMemcpyDeviceToDevice));
    RAMP_HIP_CHECK(hipMemcpy(k.wqkv_f + D * D, kk, D * D * 4, hipMemcpyDeviceToDevice));
    RAMP_HIP_CHECK(hipMemcpy(k.wqkv_f + 2 * D * D, v, D * D * 4, hipMemcpyDeviceToDevice));
    CK(permute3(c, k.wqkv_f, 3 * D, D, 1, 1, 0, 2, &k.wqkv_b));      // [256][768]
    CK(pack_linear(c, t + ".attn1.to_out.0.weight", D, D, false, &k.wo_f, &k.wo_b));
    CK(get_raw(c, t + ".attn1.to_out.0.bias", {D}, &k.bo));
    CK(pack_linear(c, t + ".ff.net.0.proj.weight", 2048, D, false, &k.w1_f, &k.w1_b));
    CK(get_raw(c, t + ".ff.net.0.proj.bias", {2048}, &k.b1));
    CK(dev_alloc(c, &k.w1_pk, 2048 * D)); CK(dev_alloc(c, &k.b1_pk, 2048));
    hipLaunchKernelGGL(geglu_pack_kernel, dim3(1024), dim3(256), 0, 0, k.w1_f, k.w1_pk, 1024, D, c->geglu_group);
    hipLaunchKernelGGL(geglu_pack_kernel, dim3(8), dim3(256), 0, 0, k.b1, k.b1_pk, 1024, 1, c->geglu_group);
    RAMP_HIP_CHECK(hipGetLastError());
    CK(pack_linear(c, t + ".ff.net.2.weight", D, 1024, false, &k.w2_f, &k.w2_b));
    CK(get_raw(c, t + ".ff.net.2.bias", {D}, &k.b2));
    CK(get_raw(c, t + ".norm1.weight", {D}, &k.ln1_g)); CK(get_raw(c, t + ".norm1.bias", {D}, &k.ln1_b));
    CK(get_raw(c, t + ".norm3.weight", {D}, &k.ln3_g)); CK(get_raw(c, t + ".norm3.bias", {D}, &k.ln3_b));
    float* tmp;
    CK(get_raw(c, t + ".attn2.to_v.weight", {D, ctx}, &tmp)); k.wv2 = tmp;
    CK(get_raw(c, t + ".attn2.to_out.0.weight", {D, D}, &tmp)); k.wo2 = tmp;
    CK(get_raw(c, t + ".attn2.to_out.0.bias", {D}, &tmp)); k.bo2 = tmp;
    // attn2.to_q / to_k / norm2 only feed a softmax over a single key (== 1): checked present, unused
    CK(get_raw(c, t + ".attn2.to_q.weight", {D, D}, &tmp));
    CK(get_raw(c, t + ".attn2.to_k.weight", {D, ctx}, &tmp));
    CK(get_raw(c, t + ".norm2.weight", {D}, &tmp)); CK(get_raw(c, t + ".norm2.bias", {D}, &tmp));
    CK(dev_alloc(c, &k.a_qkv, tok * 768)); CK(dev_alloc(c, &k.a_z1, tok * D));
    CK(dev_alloc(c, &k.a_ag, (tok + 127) * 2048)); CK(dev_alloc(c, &k.a_z2, tok * D));     // (+ 127: ffx.hip stashes whole 128-token tiles)
  }
  return 0;
}

// ---- op wrappers that count launches -----------------------------------------------------------
struct Run {
  ramp_ctx* c; hipStream_t s; int R; int row0;
  // attach the split-precision weight planes and the delayed-scaling slots of the next call site; returns 2 when the launch
  // will run the fp16x3 fragment kernels, 1 for bf16x6 fragments, 0 otherwise (< 0: error)
  int prep(GemmArgs& b) {
    if (!(c->gemm_mode >= 1 && (b.N >= 128 || (c->x6_pipe && b.N >= 64)))) return 0;
    auto it = c->x6.upper_bound(b.W);
    if (it == c->x6.begin()) return 0;
    --it;
    const auto& e = it->second;
    if (!(b.W >= it->first && b.W < it->first + e.n)) return 0;
    const size_t off = b.W - it->first;
    const bool frag = c->x6_pipe && e.packed && e.K == b.K && off % (32ul * b.K) == 0 && b.N % 32 == 0;
    if (!frag) { b.Wx = e.planes + off; b.wx_plane = (long)e.n; return 0; }
    RAMP_REQUIRE(c->site < ramp_ctx::MAX_SITES, "too many GEMM call sites for the scale table");
    int kind = 1;
    if (c->phase == 2 && e.packed3) {
      b.Wx = e.packed3 + 2 * off; b.wx_packed = 2; b.w_scale_inv = e.w_scale_inv;
      b.a_absmax_in = c->obs_in + c->site; b.a_absmax_out = c->obs_out + c->site; b.range_flag = c->range_flag;
      b.site_id = c->site;
      kind = 2;
    } else {
      b.Wx = e.packed + 3 * off; b.wx_packed = 1;
      if (c->phase >= 1) b.a_absmax_out = c->obs_out + c->site;
    }
    c->site++;
    return kind;
  }
  // a k = 5 convolution with C_in, C_out in {32, 64} on the sample-owning kernel (tkc.hip): in the calibration evaluation too (unscaled
  // operand, maximum recorded), so that the call sites number the same in every fp16x3 evaluation of a job
  const ramp_ctx::TkcW* tkc_planes(const GemmArgs& a) const {
    if (!(c->tkc_min_rows > 0 && a.M >= c->tkc_min_rows && c->gemm_mode == 2 && (c->phase == 1 || c->phase == 2) && !c->force_x6 && c->x6_pipe)) return nullptr;
    if (!(a.taps == 5 && !a.A2 && !a.Amul && !a.C2 && !a.rowbias && a.epi == EPI_LINEAR && a.a_stride == 1 && a.c_rstride == 1 && a.c_roff == 0)) return nullptr;
    if (!((a.shift0 == -2 && a.shift_step == 1) || (a.shift0 == 2 && a.shift_step == -1))) return nullptr;
    if (!tkc_applicable(a.M, a.L, a.N, a.K, nullptr)) return nullptr;
    if (!((long)a.M * a.lda * 4 < (1l << 32))) return nullptr;      // (32-bit row offsets: larger launches keep the tile kernels)
    auto it = c->tkc_w.find(a.W);
    return it == c->tkc_w.end() ? nullptr : &it->second;
  }
  struct GnPro { const float* c; const float* stats; const float* gamma; const float* beta; };
  struct GnEpi { float* cst; float* stats; const float* gamma; const float* beta; const float* tbias; float eps; };
  // ... with the GroupNorm + Mish of its Conv1dBlock fused like tkw below (pro: the input gradient's operand; epi: behind the forward convolution)
  const ramp_ctx::TkcW* tkc_gn_planes(const GemmArgs& a, bool epi) const {
    if (!(c->tkc_gn & (epi ? 1 : 2)) || (epi && a.resid2)) return nullptr;
    return tkc_planes(a);
  }
  int tkc(const GemmArgs& a, const ramp_ctx::TkcW& w, const GnPro* pro = nullptr, const GnEpi* epi = nullptr) {
    RAMP_REQUIRE(c->site < ramp_ctx::MAX_SITES, "too many GEMM call sites for the scale table");
    prof_pre(c, s, CAT_GEMM, 2.0 * a.M * a.N * a.K * a.taps, {a.M, a.N, a.K, -5});
    TkcArgs t; t.M = a.M; t.L = a.L; t.N = a.N; t.K = a.K; t.dir = a.shift_step; t.X = a.A; t.ldx = a.lda; t.W = w.planes; t.bias = a.bias;
    t.resid = a.resid; t.ldr = a.ldr; t.resid2 = a.resid2; t.ldr2 = a.ldr2; t.Y = a.C; t.ldy = a.ldc;
    t.amax_in = c->phase == 2 ? c->obs_in + c->site : nullptr; t.amax_out = c->obs_out + c->site; t.wsi = w.wsi; t.site = c->site;
    t.range_flag = c->phase == 2 ? c->range_flag : nullptr;      // (the guard judges the DELAYED scale; the calibration evaluation runs unscaled)
    if (pro) { t.gn_c = pro->c; t.gn_stats = pro->stats; t.gn_gamma = pro->gamma; t.gn_beta = pro->beta; }
    if (epi) { t.Cst = epi->cst; t.stats = epi->stats; t.gamma = epi->gamma; t.beta = epi->beta; t.tbias = epi->tbias; t.eps = epi->eps; }
    c->site++;
    int rc = launch_tkc(t, s);
    prof_post(c, s);
    c->launches++;
    return rc;
  }
  // a wide k = 5 convolution (C_out in {128, 256, 512}) with its GroupNorm fused (tkw.hip): forward = GroupNorm + Mish behind it (epi),
  // input gradient = GroupNorm backward folded into the operand (pro); consumes the call site of the tile launch it replaces
  bool use_tkw(const GemmArgs& a, bool pro, bool epi) const {
    return c->tkw_min_rows > 0 && a.M >= c->tkw_min_rows && c->gemm_mode == 2 && c->phase == 2 && !c->force_x6 && c->x6_pipe && a.taps == 5 && !a.Amul &&
           !a.rowbias && a.epi == EPI_LINEAR && a.a_stride == 1 && a.c_rstride == 1 && a.c_roff == 0 &&
           ((a.shift0 == -2 && a.shift_step == 1) || (a.shift0 == 2 && a.shift_step == -1)) && tkw_applicable(a.M, a.L, a.N, a.K, pro, epi) &&
           (!a.A2 || !pro) && (!a.C2 || !epi) && has_h3(a.W, a.N, a.K);
  }
  int tkw(const GemmArgs& a, const GnPro* pro, const GnEpi* epi) {
    GemmArgs b = a;
    const int kind = prep(b);
    if (kind < 0) return kind;
    RAMP_REQUIRE(kind == 2, "tkw: weight without fp16 fragment planes (use_tkw must have been checked)");
    prof_pre(c, s, CAT_GEMM, 2.0 * a.M * a.N * a.K * 5, {a.M, a.N, a.K, -7});
    TkwArgs t; t.M = a.M; t.L = a.L; t.N = a.N; t.K = a.K; t.dir = a.shift_step; t.X = a.A; t.ldx = a.lda; t.X2 = a.A2; t.ldx2 = a.lda2;
    t.K1 = a.A2 ? a.K1 : a.K; t.W = b.Wx; t.wsi = b.w_scale_inv; t.bias = a.bias; t.resid = a.resid; t.ldr = a.ldr; t.resid2 = a.resid2; t.ldr2 = a.ldr2;
    t.Y = a.C; t.ldy = a.ldc; t.Y2 = a.C2; t.ldy2 = a.ldc2; t.N1 = a.C2 ? a.N1 : a.N;
    if (pro) { t.gn_c = pro->c; t.gn_stats = pro->stats; t.gn_gamma = pro->gamma; t.gn_beta = pro->beta; }
    if (epi) { t.Cst = epi->cst; t.stats = epi->stats; t.gamma = epi->gamma; t.beta = epi->beta; t.tbias = epi->tbias; t.eps = epi->eps; }
    t.amax_in = b.a_absmax_in; t.amax_out = b.a_absmax_out; t.site = b.site_id; t.range_flag = b.range_flag;
    int rc = launch_tkw(t, s);
    prof_post(c, s);
    c->launches++;
    return rc;
  }
  int gemm(const GemmArgs& a) {
    if (const ramp_ctx::TkcW* w = tkc_planes(a)) return tkc(a, *w);
    prof_pre(c, s, CAT_GEMM, 2.0 * a.M * a.N * a.K * a.taps, {a.M, a.N, a.K, a.taps});
    GemmArgs b = a;
    b.three_ok = c->three_blocks;
    const int kind = prep(b);
    if (kind < 0) return kind;
    int rc = launch_gemm(b, s);
    prof_post(c, s);
    c->launches++;
    return rc;
  }
  // the token-owning fused feed-forward (ffx.hip) serves this feed-forward pair, forward AND backward (the stash layout is
  // private to the two kernels, so both directions of an evaluation must agree: same phase, same M)
  bool use_ffx(const STBlock& k, int M) const {
    return k.ffx_f && c->ffx_min_rows > 0 && M >= c->ffx_min_rows && c->gemm_mode == 2 && c->phase == 2 && c->x6_pipe;
  }
  // consumes the two call sites of the launches it replaces (forward: FF1, FF2; backward: d(hg), FF1-dX), in their order
  int ffx(const STBlock& k, bool bwd, const float* X, const float* z1, float* Y, int M) {
    RAMP_REQUIRE(c->site + 2 <= ramp_ctx::MAX_SITES, "too many GEMM call sites for the scale table");
    prof_pre(c, s, CAT_GEMM, 2.0 * M * (2048.0 * 256 + 256.0 * 1024), {M, bwd ? -3 : -2, 256, 2});
    FfxArgs f; f.M = M; f.X = X; f.Z1 = z1; f.Y = Y; f.stash = k.a_ag; f.ln_g = k.ln3_g; f.ln_b = k.ln3_b;
    const bool s16 = c->mfma16 && k.ffx16_f;
    f.Wstream = s16 ? (bwd ? k.ffx16_b : k.ffx16_f) : (bwd ? k.ffx_b : k.ffx_f); f.b1 = k.b1_pk; f.b2 = k.b2; f.range_flag = c->range_flag;
    f.amax_in1 = c->obs_in + c->site; f.amax_out1 = c->obs_out + c->site; f.site1 = c->site;
    f.amax_in2 = c->obs_in + c->site + 1; f.amax_out2 = c->obs_out + c->site + 1; f.site2 = c->site + 1;
    f.wsi1 = bwd ? k.ffx_wsi_w2 : k.ffx_wsi_w1; f.wsi2 = bwd ? k.ffx_wsi_w1 : k.ffx_wsi_w2;
    f.ablate = s16 ? 0 : c->ffx_ablate;
    c->site += 2;
    int rc = s16 ? launch_ffx16(f, bwd, s) : launch_ffx(f, bwd, s);
    prof_post(c, s);
    c->launches++;
    return rc;
  }
  // a K = 256 linear on the token-owning kernel (tkl.hip), optionally with LayerNorm folded into its operand; consumes the
  // call site of the tile-kernel launch it replaces (same operand, same maxima)
  // the weight W [N][K] carries fragment-packed fp16 planes (what prep() would attach in an fp16x3 evaluation): a weight without
  // them keeps its launch on the tile kernels instead of failing inside the token-owning wrapper
  bool has_h3(const float* W, int N, int K) const {
    auto it = c->x6.upper_bound(W);
    if (it == c->x6.begin()) return false;
    --it;
    const auto& e = it->second;
    if (!(W >= it->first && W < it->first + e.n)) return false;
    const size_t off = W - it->first;
    return c->x6_pipe && e.packed && e.packed3 && e.K == K && off % (32ul * K) == 0 && N % 32 == 0;
  }
  bool use_tkl(const GemmArgs& a) const {
    return c->tkl_min_rows > 0 && a.M >= c->tkl_min_rows && c->gemm_mode == 2 && c->phase == 2 && c->x6_pipe && a.K == 256 && a.lda == 256 &&
           a.taps == 1 && a.N % 32 == 0 && a.N <= 768 && !a.A2 && !a.Amul && !a.resid2 && !a.C2 && a.epi == EPI_LINEAR && has_h3(a.W, a.N, a.K);
  }
  int tkl(const GemmArgs& a, const float* ln_g, const float* ln_b) {
    GemmArgs b = a;
    const int kind = prep(b);
    if (kind < 0) return kind;
    RAMP_REQUIRE(kind == 2, "tkl: weight without fp16 fragment planes (use_tkl must have been checked)");
    prof_pre(c, s, CAT_GEMM, 2.0 * a.M * a.N * a.K, {a.M, a.N, a.K, ln_g ? -11 : -10});
    TklArgs t; t.M = a.M; t.N = a.N; t.X = a.A; t.Y = a.C; t.ldy = a.ldc; t.W = b.Wx; t.bias = a.bias; t.resid = a.resid; t.ldr = a.ldr;
    t.rowbias = a.rowbias; t.rowvar = a.rowvar; t.row0 = a.row0; t.rb_stride = a.rb_stride; t.L = a.L; t.n_var = a.rowbias ? c->n_variants : 0;
    t.ln_g = ln_g; t.ln_b = ln_b; t.amax_in = b.a_absmax_in; t.amax_out = b.a_absmax_out; t.wsi = b.w_scale_inv; t.site = b.site_id;
    t.range_flag = b.range_flag;
    auto w16 = c->mfma16 ? c->tk16_w.find(a.W) : c->tk16_w.end();      // the same linear on v_mfma_f32_16x16x32_f16 (tkl16.hip): its own fragment planes, same scale
    if (w16 != c->tk16_w.end()) t.W = w16->second;
    int rc = w16 != c->tk16_w.end() ? launch_tkl16(t, s) : launch_tkl(t, s);
    prof_post(c, s);
    c->launches++;
    return rc;
  }
  // self-attention + output projection (+ bias, + the row variant's cross-attention constant, + residual) in one launch of
  // sample-owning waves (atk.hip); consumes the call site of the out-projection launch it replaces (same operand o, same maxima)
  bool use_ato(const STBlock& k, int M, int L) const {
    return k.ato_w && c->atk_min_rows > 0 && M >= c->atk_min_rows && c->gemm_mode == 2 && c->phase == 2 && c->x6_pipe && c->n_variants <= 4 &&
           ato_applicable(M, L, nullptr);
  }
  bool use_atb(int M, int L) const {
    return c->atk_min_rows > 0 && M >= c->atk_min_rows && c->gemm_mode == 2 && c->phase == 2 && c->x6_pipe && ato_applicable(M, L, nullptr);
  }
  int ato(const STBlock& k, const float* qkv, const float* resid, float* Y, int M, int L, const float* rowbias, int rb_stride) {
    RAMP_REQUIRE(c->site < ramp_ctx::MAX_SITES, "too many GEMM call sites for the scale table");
    prof_pre(c, s, CAT_GEMM, 2.0 * M * 256 * 256 + 16.0 * M * L * 64, {M, 256, 256, -4});
    AtoArgs t; t.M = M; t.L = L; t.QKV = qkv; t.W = k.ato_w; t.bias = k.bo; t.resid = resid; t.Y = Y;
    t.rowbias = rowbias; t.rowvar = c->row_variant; t.row0 = row0; t.rb_stride = rb_stride; t.n_var = rowbias ? c->n_variants : 0;
    t.amax_in = c->obs_in + c->site; t.amax_out = c->obs_out + c->site; t.wsi = k.ato_wsi; t.site = c->site; t.range_flag = c->range_flag;
    c->site++;
    int rc = launch_ato(t, s);
    prof_post(c, s);
    c->launches++;
    return rc;
  }
  // attention backward + d(ln1) + LayerNorm-1 backward in one launch of sample-owning waves (atl.hip): d(qkv) never reaches HBM;
  // consumes the call site of the d(ln1) GEMM it contains
  bool use_abl(const STBlock& k, int M, int L) const {
    return c->abl_on && k.abl_w && use_atb(M, L) && !c->tklb_off && c->tkl_min_rows > 0 && has_h3(k.wqkv_b, 256, 768);      // (from atk_rows tokens on)
  }
  int abl(const STBlock& k, const float* qkv, const float* dout, const float* z, const float* add, float* out, int M, int L) {
    GemmArgs b; b.A = c->t_dqkv; b.lda = 768; b.W = k.wqkv_b; b.C = out; b.ldc = 256; b.M = M; b.N = 256; b.K = 768; b.taps = 1; b.L = 1;
    const int kind = prep(b);                               // (the site's scale slots and the weight's scale; no operand is read through b)
    if (kind < 0) return kind;
    RAMP_REQUIRE(kind == 2, "abl: weight without fp16 fragment planes (use_abl must have been checked)");
    prof_pre(c, s, CAT_GEMM, 2.0 * M * 256 * 768 + 32.0 * M * L * 64, {M, 256, 768, -6});
    AblArgs t; t.M = M; t.L = L; t.QKV = qkv; t.dO = dout; t.W = k.abl_w; t.Z = z; t.add = add; t.ln_g = k.ln1_g; t.Y = out;
    t.amax_in = b.a_absmax_in; t.amax_out = b.a_absmax_out; t.wsi = b.w_scale_inv; t.site = b.site_id; t.range_flag = b.range_flag;
    int rc = launch_abl(t, s);
    prof_post(c, s);
    c->launches++;
    return rc;
  }
  // d(ln1) = d(qkv) Wqkv^T and the LayerNorm-1 backward behind it in one token-owning launch (tkl.hip, tklb_kernel); consumes
  // the call site of the d(ln1) GEMM it replaces
  bool use_tklb(int M, const float* W) const {
    return !c->tklb_off && c->tkl_min_rows > 0 && M >= c->tkl_min_rows && c->gemm_mode == 2 && c->phase == 2 && c->x6_pipe && has_h3(W, 256, 768);
  }
  int tklb(const float* dqkv, const float* W, const float* z, const float* ln_g, const float* add, float* out, int M) {
    GemmArgs b; b.A = dqkv; b.lda = 768; b.W = W; b.C = out; b.ldc = 256; b.M = M; b.N = 256; b.K = 768; b.taps = 1; b.L = 1;
    const int kind = prep(b);
    if (kind < 0) return kind;
    RAMP_REQUIRE(kind == 2, "tklb: weight without fp16 fragment planes (use_tklb must have been checked)");
    prof_pre(c, s, CAT_GEMM, 2.0 * M * 256 * 768, {M, 256, 768, -12});
    TklbArgs t; t.M = M; t.X = dqkv; t.Z = z; t.add = add; t.Y = out; t.W = b.Wx; t.ln_g = ln_g;
    t.amax_in = b.a_absmax_in; t.amax_out = b.a_absmax_out; t.wsi = b.w_scale_inv; t.site = b.site_id; t.range_flag = b.range_flag;
    int rc = launch_tklb(t, s);
    prof_post(c, s);
    c->launches++;
    return rc;
  }
  // FF1 -> GEGLU -> FF2 of one transformer block (layers_attention_mini.py:38-45, 147): one fused launch in the fp16x3
  // evaluations (the 1024-wide hidden stays in LDS), two launches otherwise.  Either way the two call sites are
  // numbered in the same order, so calibration and fused evaluations read each other's maxima.
  int ff_forward(const GemmArgs& u, const GemmArgs& f2) {
    GemmArgs b1 = u, b2 = f2;
    if (c->ff_fused && u.M >= c->ff_fused && c->gemm_mode == 2 && c->phase == 2 && c->x6_pipe) {
      prof_pre(c, s, CAT_GEMM, 2.0 * u.M * u.N * u.K + 2.0 * f2.M * f2.N * f2.K, {u.M, -1, u.K, 2});
      const int k1 = prep(b1); if (k1 < 0) return k1;
      const int k2 = prep(b2); if (k2 < 0) return k2;
      RAMP_REQUIRE(k1 == 2 && k2 == 2, "fused feed-forward: weights without fp16 planes");
      int rc = launch_ff_fwd(b1, b2, s);
      prof_post(c, s);
      c->launches++;
      return rc;
    }
    if (int rc = gemm(u)) return rc;
    return gemm(f2);
  }
};

GemmArgs lin(const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int M, int N, int K) {
  GemmArgs a; a.A = A; a.lda = lda; a.W = W; a.bias = bias; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
  a.taps = 1; a.shift0 = 0; a.shift_step = 0; a.L = 1;
  return a;
}
GemmArgs conv5(const float* A, int lda, const float* W, const float* bias, float* C, int ldc, int M, int N, int K,
               int L, bool backward) {
  GemmArgs a = lin(A, lda, W, bias, C, ldc, M, N, K);
  a.taps = 5; a.L = L;
  if (backward) { a.shift0 = 2; a.shift_step = -1; } else { a.shift0 = -2; a.shift_step = 1; }
  return a;
}

int dbg_store(ramp_ctx* c, const std::string& key, const float* src, size_t n, hipStream_t s) {
  if (!c->cfg.debug_taps) return 0;
  auto it = c->dbg.find(key);
  if (it == c->dbg.end() || it->second.second < n) {
    float* p; CK(dev_alloc(c, &p, n));
    c->dbg[key] = {p, n};
    it = c->dbg.find(key);
  }
  RAMP_HIP_CHECK(hipMemcpyAsync(it->second.first, src, n * 4, hipMemcpyDeviceToDevice, s));
  return 0;
}

// ---- ResidualTemporalBlock ---------------------------------------------------------------------
// xa (R,L,Ca) [, xb (R,L,Cb) concatenated along channels]; x_first: (B,H,S) for the first layer
int rtb_forward(Run& r, RTB& m, const float* xa, int ca, const float* xb, int cb, const float* x_first, int n_rp,
                int t) {
  ramp_ctx* c = r.c; const int R = r.R, M = R * m.L;
  const float* tbias = c->time_table + (size_t)t * c->tt_stride + m.tb_off;
  const float* resid;
  bool fused1 = false;
  if (m.first) {
    LAUNCH(c, r.s, CAT_SMALLCONV, 0, launch_conv_in_fwd(x_first, m.w5in, m.c1.bias, m.w1in, m.res_bias, m.a_c1, c->t_res, R, n_rp, m.L, m.cin, r.s));
    resid = c->t_res;
  } else {
    GemmArgs a = conv5(xa, ca, m.c1.fwd, m.c1.bias, m.a_c1, m.cout, M, m.cout, m.cin, m.L, false);
    if (xb) { a.A2 = xb; a.lda2 = cb; a.K1 = ca; }
    fused1 = r.use_tkw(a, false, true);
    if (fused1) {      // conv -> (stash c1, statistics) -> GroupNorm -> Mish -> + time bias = h, one launch (tkw.hip)
      Run::GnEpi e{m.a_c1, m.a_st1, m.g1, m.b1, tbias, 1e-5f};
      a.C = m.a_h;
      CK(r.tkw(a, nullptr, &e));
    } else if (const ramp_ctx::TkcW* w = r.tkc_gn_planes(a, true)) {      // the same on the narrow levels (tkc.hip)
      Run::GnEpi e{m.a_c1, m.a_st1, m.g1, m.b1, tbias, 1e-5f};
      a.C = m.a_h; a.ldc = m.cout;
      CK(r.tkc(a, *w, nullptr, &e));
      fused1 = true;
    } else
    CK(r.gemm(a));
    if (m.has_res) {
      GemmArgs b = lin(xa, ca, m.res_f, m.res_bias, c->t_res, m.cout, M, m.cout, m.cin);
      if (xb) { b.A2 = xb; b.lda2 = cb; b.K1 = ca; }
      CK(r.gemm(b));
      resid = c->t_res;
    } else {
      resid = xa;
    }
  }
  GnArgs g; g.x = m.a_c1; g.gamma = m.g1; g.beta = m.b1; g.tbias = tbias; g.resid = nullptr; g.y = m.a_h;
  g.stats = m.a_st1; g.R = R; g.L = m.L; g.C = m.cout; g.eps = 1e-5f; g.mish = 1;
  if (!fused1) LAUNCH(c, r.s, CAT_ROW, 0, launch_gn_fwd(g, r.s));
  GemmArgs c2a = conv5(m.a_h, m.cout, m.c2.fwd, m.c2.bias, m.a_c2, m.cout, M, m.cout, m.cout, m.L, false);
  if (r.use_tkw(c2a, false, true)) {      // conv -> (stash c2, statistics) -> GroupNorm -> Mish -> + residual = the block's output
    Run::GnEpi e{m.a_c2, m.a_st2, m.g2, m.b2, nullptr, 1e-5f};
    c2a.C = m.a_out; c2a.resid = resid; c2a.ldr = m.cout;
    CK(r.tkw(c2a, nullptr, &e));
  } else if (const ramp_ctx::TkcW* w = r.tkc_gn_planes(c2a, true)) {
    Run::GnEpi e{m.a_c2, m.a_st2, m.g2, m.b2, nullptr, 1e-5f};
    c2a.C = m.a_out; c2a.resid = resid; c2a.ldr = m.cout;
    CK(r.tkc(c2a, *w, nullptr, &e));
  } else {
  CK(r.gemm(c2a));
  g.x = m.a_c2; g.gamma = m.g2; g.beta = m.b2; g.tbias = nullptr; g.resid = resid; g.y = m.a_out; g.stats = m.a_st2;
  LAUNCH(c, r.s, CAT_ROW, 0, launch_gn_fwd(g, r.s));
  }
  CK(dbg_store(c, "out/" + m.name, m.a_out, (size_t)M * m.cout, r.s));
  return 0;
}

// dy (R,L,Cout) -> dxa (R,L,Ca) [, dxb (R,L,Cb)] ; add2: optional extra addend on dxa (same shape)
int rtb_backward(Run& r, RTB& m, const float* dy, float* dxa, int ca, float* dxb, int cb, const float* add2,
                 float* eps_out) {
  ramp_ctx* c = r.c; const int R = r.R, M = R * m.L;
  CK(dbg_store(c, "gout/" + m.name, dy, (size_t)M * m.cout, r.s));
  GnBwdArgs g; g.dy = dy; g.x = m.a_c2; g.stats = m.a_st2; g.gamma = m.g2; g.beta = m.b2; g.add = nullptr;
  g.dx = c->g_t1; g.R = R; g.L = m.L; g.C = m.cout; g.mish = 1;
  GemmArgs c2b = conv5(c->g_t1, m.cout, m.c2.bwd, nullptr, c->g_t2, m.cout, M, m.cout, m.cout, m.L, true);
  if (r.use_tkw(c2b, true, false)) {      // dh = conv2^T(GNbwd(dy mish'; c2)): the GroupNorm backward is the operand staging of the convolution (tkw.hip)
    Run::GnPro p{m.a_c2, m.a_st2, m.g2, m.b2};
    c2b.A = dy; c2b.lda = m.cout;
    CK(r.tkw(c2b, &p, nullptr));
  } else if (const ramp_ctx::TkcW* w = r.tkc_gn_planes(c2b, false)) {
    Run::GnPro p{m.a_c2, m.a_st2, m.g2, m.b2};
    c2b.A = dy; c2b.lda = m.cout;
    CK(r.tkc(c2b, *w, &p, nullptr));
  } else {
  LAUNCH(c, r.s, CAT_ROW, 0, launch_gn_bwd(g, r.s));                                                     // dc2
  CK(r.gemm(c2b));                                                                                       // dh
  }
  g.dy = c->g_t2; g.x = m.a_c1; g.stats = m.a_st1; g.gamma = m.g1; g.beta = m.b1; g.dx = c->g_t1;
  if (m.first) {
    LAUNCH(c, r.s, CAT_ROW, 0, launch_gn_bwd(g, r.s));                                                   // dc1
    LAUNCH(c, r.s, CAT_SMALLCONV, 0, launch_conv_in_bwd(c->g_t1, dy, m.w5in, m.w1in, eps_out, R, m.L, m.cin, r.s));
    return 0;
  }
  const float* resid; int ldr;
  if (m.has_res) {
    CK(r.gemm(lin(dy, m.cout, m.res_b, nullptr, c->g_tr, m.cin, M, m.cin, m.cout)));
    resid = c->g_tr; ldr = m.cin;
  } else {
    resid = dy; ldr = m.cout;
  }
  GemmArgs a = conv5(c->g_t1, m.cout, m.c1.bwd, nullptr, dxa, ca, M, m.cin, m.cout, m.L, true);
  a.resid = resid; a.ldr = ldr;
  if (dxb) { a.C2 = dxb; a.ldc2 = cb; a.N1 = ca; }
  if (add2) { RAMP_REQUIRE(dxb == nullptr, "add2 with split output"); a.resid2 = add2; a.ldr2 = ca; }
  if (r.use_tkw(a, true, false)) {        // dx = conv1^T(GNbwd(dh mish'; c1)) + residual path (+ skip gradient), one launch
    Run::GnPro p{m.a_c1, m.a_st1, m.g1, m.b1};
    a.A = c->g_t2; a.lda = m.cout;
    CK(r.tkw(a, &p, nullptr));
    return 0;
  }
  if (const ramp_ctx::TkcW* w = r.tkc_gn_planes(a, false)) {
    Run::GnPro p{m.a_c1, m.a_st1, m.g1, m.b1};
    a.A = c->g_t2; a.lda = m.cout;
    return r.tkc(a, *w, &p, nullptr);
  }
  LAUNCH(c, r.s, CAT_ROW, 0, launch_gn_bwd(g, r.s));                                                     // dc1
  CK(r.gemm(a));
  return 0;
}

// ---- SpatialTransformer ------------------------------------------------------------------------
// share > 1 (first transformer of the network inside a sampling job): x and everything up to the first block's
// attention output projection hold ONE row per trajectory (r.R / share rows); the rows part ways where the
// cross-attention constant of their scene variant is added (expand_rows), and the outer residual uses x expanded.
int st_forward(Run& r, ST& m, const float* x, int share = 1) {
  ramp_ctx* c = r.c; const int R = r.R, M = R * m.L, D = 256;
  const int Rp = R / share, Mp = Rp * m.L;              // prefix rows
  GnArgs g; g.x = x; g.gamma = m.gn_g; g.beta = m.gn_b; g.y = c->t_xn; g.stats = m.a_gst; g.R = Rp; g.L = m.L;
  g.C = m.C; g.eps = 1e-6f; g.mish = 0;
  LAUNCH(c, r.s, CAT_ROW, 0, launch_gn_fwd(g, r.s));
  CK(r.gemm(lin(c->t_xn, m.C, m.wpi_f, m.bpi, m.a_z0, D, Mp, D, m.C)));
  const float* zin = m.a_z0;
  for (int b = 0; b < 2; ++b) {
    STBlock& k = m.blk[b];
    const bool pre = share > 1 && b == 0;
    const int Rb = pre ? Rp : R, Mb = Rb * m.L;
    {
      GemmArgs q = lin(c->t_ln, D, k.wqkv_f, nullptr, k.a_qkv, 768, Mb, 768, D);
      if (r.use_tkl(q)) {      // LayerNorm-1 on the operand registers of the token-owning QKV kernel: no ln_fwd, no t_ln round trip
        q.A = zin;
        CK(r.tkl(q, k.ln1_g, k.ln1_b));
      } else {
        LAUNCH(c, r.s, CAT_ROW, 0, launch_ln_fwd(zin, k.ln1_g, k.ln1_b, c->t_ln, Mb, r.s));
        CK(r.gemm(q));
      }
    }
    const float* rowbias = c->cross_bias + (size_t)(m.blk0 + b) * D; const int rb_stride = c->n_blocks_total * D;
    if (r.use_ato(k, Mb, m.L)) {      // softmax(q k^T / 8) v -> out-projection + bias + constant + residual: o never reaches HBM
      CK(r.ato(k, k.a_qkv, zin, pre ? c->t_ln : k.a_z1, Mb, m.L, pre ? nullptr : rowbias, rb_stride));
    } else {
    LAUNCH(c, r.s, CAT_ATTN, 16.0 * Rb * m.L * m.L * 64, launch_attn_fwd(k.a_qkv, c->t_o, Rb, m.L, r.s));
    GemmArgs a = lin(c->t_o, D, k.wo_f, k.bo, pre ? c->t_ln : k.a_z1, D, Mb, D, D);
    a.resid = zin; a.ldr = D; a.L = m.L;
    if (!pre) { a.rowbias = rowbias; a.rb_stride = rb_stride; a.rowvar = c->row_variant; a.row0 = r.row0; }
    if (r.use_tkl(a) && (!a.rowbias || c->n_variants <= 4)) CK(r.tkl(a, nullptr, nullptr)); else CK(r.gemm(a));
    }
    if (pre) LAUNCH(c, r.s, CAT_ROW, 0, launch_expand_rows(c->t_ln, k.a_z1, R, share, m.L, D, rowbias, rb_stride, c->row_variant, r.row0, r.s));
    if (r.use_ffx(k, M)) {      // LN3 -> FF1 -> GEGLU -> FF2 -> + z1 in one launch, nothing but the stash and z2 written
      CK(r.ffx(k, false, k.a_z1, k.a_z1, k.a_z2, M));
      zin = k.a_z2;
      continue;
    }
    LAUNCH(c, r.s, CAT_ROW, 0, launch_ln_fwd(k.a_z1, k.ln3_g, k.ln3_b, c->t_ln, M, r.s));
    GemmArgs u = lin(c->t_ln, D, k.w1_pk, k.b1_pk, k.a_ag, 2048, M, 2048, D);
    u.epi = EPI_GEGLU_FWD; u.aux_out = c->t_hg; u.ld_aux = 1024;        // writes ag (stash) and hg = a * gelu(g)
    u.geglu_group = c->geglu_group;
    GemmArgs f = lin(c->t_hg, 1024, k.w2_f, k.b2, k.a_z2, D, M, D, 1024);
    f.resid = k.a_z1; f.ldr = D;
    CK(r.ff_forward(u, f));
    zin = k.a_z2;
  }
  GemmArgs o = lin(zin, D, m.wpo_f, m.bpo, m.a_y, m.C, M, m.C, D);
  o.resid = x; o.ldr = m.C;
  if (share > 1) {
    LAUNCH(c, r.s, CAT_ROW, 0, launch_expand_rows(x, c->t_xn, R, share, m.L, m.C, nullptr, 0, nullptr, 0, r.s));
    o.resid = c->t_xn;
  }
  CK(r.gemm(o));
  CK(dbg_store(c, "out/" + m.name, m.a_y, (size_t)M * m.C, r.s));
  return 0;
}

// share > 1: see st_forward; comb = the weights of the rows' gradients in what the sampler uses.  dy, and everything down
// to d(z1) of the first block, have r.R rows; from there on (and dx) one COMBINED row per trajectory.
int st_backward(Run& r, ST& m, const float* x, const float* dy, float* dx, int share = 1, const float* comb = nullptr) {
  ramp_ctx* c = r.c; const int R = r.R, M = R * m.L, D = 256;
  const int Rp = R / share, Mp = Rp * m.L;
  CK(dbg_store(c, "gout/" + m.name, dy, (size_t)M * m.C, r.s));
  float* dz = c->t_dz; float* dz1 = c->t_dz1;
  CK(r.gemm(lin(dy, m.C, m.wpo_b, nullptr, dz, D, M, D, m.C)));
  for (int b = 1; b >= 0; --b) {
    STBlock& k = m.blk[b];
    const float* zin = (b == 0) ? m.a_z0 : m.blk[0].a_z2;
    if (r.use_ffx(k, M)) {      // dz1 = dz + LN3bwd(W1^T (d(hg) (.) stash)), d(hg) = W2^T dz: one launch
      CK(r.ffx(k, true, dz, k.a_z1, dz1, M));
    } else {
    if (c->gemm_mode >= 1 && c->x6_pipe) {
      // d(hg) only (1024 wide); d(ag) = [d(hg) s1 | d(hg) s2] is formed by the next GEMM's operand loader from the
      // forward stash: the 2048-wide d(ag) never goes to HBM (saves a 2048-float write and a 2048-float read per token)
      CK(r.gemm(lin(dz, D, k.w2_b, nullptr, c->t_hg, 1024, M, 1024, D)));
      GemmArgs v = lin(c->t_hg, 1024, k.w1_b, nullptr, c->t_dln, D, M, D, 2048);         // d(ln3)
      v.Amul = k.a_ag; v.lda_mul = 2048; v.a_period = 1024;
      CK(r.gemm(v));
    } else {
      GemmArgs w = lin(dz, D, k.w2_b, nullptr, c->t_dag, 2048, M, 1024, D);   // d(hg) -> d(ag) in the epilogue
      w.epi = EPI_GEGLU_BWD; w.aux_in = k.a_ag; w.ld_aux = 2048;
      CK(r.gemm(w));
      CK(r.gemm(lin(c->t_dag, 2048, k.w1_b, nullptr, c->t_dln, D, M, D, 2048)));        // d(ln3)
    }
    LAUNCH(c, r.s, CAT_ROW, 0, launch_ln_bwd(c->t_dln, k.a_z1, k.ln3_g, dz, dz1, M, r.s));         // dz1
    }
    const bool pre = share > 1 && b == 0;
    const int Rb = pre ? Rp : R, Mb = Rb * m.L;
    if (pre) {      // the rows of a trajectory meet again: dz (free now) <- sum_j comb_j dz1[row j]
      LAUNCH(c, r.s, CAT_ROW, 0, launch_combine_rows(dz1, dz, Rp, share, m.L, D, comb, r.s));
      std::swap(dz, dz1);
    }
    {
      GemmArgs o = lin(dz1, D, k.wo_b, nullptr, c->t_o, D, Mb, D, D);                   // d(o)
      if (r.use_tkl(o)) CK(r.tkl(o, nullptr, nullptr)); else CK(r.gemm(o));
    }
    if (r.use_abl(k, Mb, m.L)) {   // d(q, k, v), d(ln1) and the LayerNorm-1 backward in ONE launch: dz = dz1 + LN1bwd(attention-backward(d(o)) Wqkv^T)
      CK(r.abl(k, k.a_qkv, c->t_o, zin, dz1, dz, Mb, m.L));
    } else {
    if (r.use_atb(Mb, m.L)) {      // d(q, k, v) on sample-owning waves (atk.hip, atb_kernel): fp16x3 MFMAs, no LDS tile, two waves per SIMD
      AtbArgs t; t.M = Mb; t.L = m.L; t.QKV = k.a_qkv; t.dO = c->t_o; t.dQKV = c->t_dqkv;
      LAUNCH(c, r.s, CAT_ATTN, 32.0 * Rb * m.L * m.L * 64, launch_atb(t, r.s));
    } else
    LAUNCH(c, r.s, CAT_ATTN, 32.0 * Rb * m.L * m.L * 64, launch_attn_bwd(k.a_qkv, c->t_o, c->t_dqkv, Rb, m.L, r.s));
    if (r.use_tklb(Mb, k.wqkv_b)) {      // d(ln1) and the LayerNorm-1 backward in one token-owning launch: dz = dz1 + LN1bwd(d(qkv) Wqkv^T)
      CK(r.tklb(c->t_dqkv, k.wqkv_b, zin, k.ln1_g, dz1, dz, Mb));
    } else {
      CK(r.gemm(lin(c->t_dqkv, 768, k.wqkv_b, nullptr, c->t_dln, D, Mb, D, 768)));        // d(ln1)
      LAUNCH(c, r.s, CAT_ROW, 0, launch_ln_bwd(c->t_dln, zin, k.ln1_g, dz1, dz, Mb, r.s));           // dz (block input)
    }
    }
  }
  CK(r.gemm(lin(dz, D, m.wpi_b, nullptr, c->t_xn, m.C, Mp, m.C, D)));                   // d(xn)
  GnBwdArgs g; g.dy = c->t_xn; g.x = x; g.stats = m.a_gst; g.gamma = m.gn_g; g.beta = m.gn_b; g.add = dy; g.dx = dx;
  g.R = Rp; g.L = m.L; g.C = m.C; g.mish = 0;
  if (share > 1) {   // the gradient through the outer residual, combined (t_o is free after the attention backward)
    LAUNCH(c, r.s, CAT_ROW, 0, launch_combine_rows(dy, c->t_o, Rp, share, m.L, m.C, comb, r.s));
    g.add = c->t_o;
  }
  LAUNCH(c, r.s, CAT_ROW, 0, launch_gn_bwd(g, r.s));
  return 0;
}

// ---- whole network -----------------------------------------------------------------------------
// x (B,H,S) device pointer of the FIRST trajectory of this chunk; rows [row0, row0 + R)
// share = n_rp (> 1) inside a sampling job: the rows of a trajectory carry the same x and t, so the two residual blocks
// of level 0 and the first transformer down to its first cross-attention constant run on one row per trajectory
int net_forward(ramp_ctx* c, const float* x_chunk, int row0, int R, int n_rp, int t, float* f_out, bool want_grad,
                hipStream_t s, int share = 1) {
  Run r{c, s, R, row0};
  Run rp{c, s, R / share, row0 / share};
  const int nl = c->cfg.n_levels;
  const float* cur = nullptr; int cc = 0;
  for (int k = 0; k < nl; ++k) {
    RTB& a = c->rtbs[2 * k]; RTB& b = c->rtbs[2 * k + 1]; ST& st = c->sts[k];
    const bool pre = share > 1 && k == 0;
    CK(rtb_forward(pre ? rp : r, a, cur, cc, nullptr, 0, x_chunk, pre ? 1 : n_rp, t));
    CK(rtb_forward(pre ? rp : r, b, a.a_out, a.cout, nullptr, 0, nullptr, pre ? 1 : n_rp, t));
    CK(st_forward(r, st, b.a_out, pre ? share : 1));
    if (k < nl - 1) {
      Resample& d = c->downs[k];
      // Downsample1d: y[lo] = b + sum_j x[2 lo + j - 1] W_j^T  -> 3 taps over stride-2 source rows
      GemmArgs ga = lin(st.a_y, d.C, d.w_f, d.bias, d.a_y, d.C, R * d.Lout, d.C, d.C);
      ga.taps = 3; ga.shift0 = -1; ga.shift_step = 1; ga.L = d.Lout; ga.a_stride = 2;
      CK(r.gemm(ga));
      CK(dbg_store(c, "out/" + d.name, d.a_y, (size_t)R * d.Lout * d.C, s));
      cur = d.a_y; cc = d.C;
    } else {
      cur = st.a_y; cc = st.C;
    }
  }
  RTB& m1 = c->rtbs[2 * nl]; RTB& m2 = c->rtbs[2 * nl + 1]; ST& ms = c->sts[nl];
  CK(rtb_forward(r, m1, cur, cc, nullptr, 0, nullptr, n_rp, t));
  CK(st_forward(r, ms, m1.a_out));
  CK(rtb_forward(r, m2, ms.a_y, ms.C, nullptr, 0, nullptr, n_rp, t));
  cur = m2.a_out; cc = m2.cout;
  for (int k = 0; k < nl - 1; ++k) {
    RTB& a = c->rtbs[2 * nl + 2 + 2 * k]; RTB& b = c->rtbs[2 * nl + 3 + 2 * k]; ST& st = c->sts[nl + 1 + k];
    ST& skip = c->sts[nl - 1 - k];
    CK(rtb_forward(r, a, cur, cc, skip.a_y, skip.C, nullptr, n_rp, t));
    CK(rtb_forward(r, b, a.a_out, a.cout, nullptr, 0, nullptr, n_rp, t));
    CK(st_forward(r, st, b.a_out));
    Resample& u = c->ups[k];
    // Upsample1d (ConvTranspose1d k4 s2 p1): y[2i] = b + x[i] W_1 + x[i-1] W_3 ; y[2i+1] = b + x[i+1] W_0 + x[i] W_2
    for (int par = 0; par < 2; ++par) {
      GemmArgs ga = lin(st.a_y, u.C, u.w_f + (size_t)par * 2 * u.C * u.C, u.bias, u.a_y, u.C, R * u.Lin, u.C, u.C);
      ga.taps = 2; ga.shift0 = par; ga.shift_step = -1; ga.L = u.Lin; ga.c_rstride = 2; ga.c_roff = par;
      CK(r.gemm(ga));
    }
    CK(dbg_store(c, "out/" + u.name, u.a_y, (size_t)R * u.Lout * u.C, s));
    cur = u.a_y; cc = u.C;
  }
  const int H = c->cfg.horizon, C0 = c->cfg.unet_input_dim, M = R * H;
  GemmArgs fa = conv5(cur, cc, c->final_conv.fwd, c->final_conv.bias, c->a_fin_c, C0, M, C0, C0, H, false);
  if (const ramp_ctx::TkcW* w = r.tkc_gn_planes(fa, true)) {
    Run::GnEpi e{c->a_fin_c, c->a_fin_st, c->fin_g, c->fin_b, nullptr, 1e-5f};
    fa.C = c->a_fin_a; fa.ldc = C0;
    CK(r.tkc(fa, *w, nullptr, &e));
  } else {
  CK(r.gemm(fa));
  GnArgs g; g.x = c->a_fin_c; g.gamma = c->fin_g; g.beta = c->fin_b; g.y = c->a_fin_a; g.stats = c->a_fin_st;
  g.R = R; g.L = H; g.C = C0; g.eps = 1e-5f; g.mish = 1;
  LAUNCH(c, s, CAT_ROW, 0, launch_gn_fwd(g, s));
  }
  LAUNCH(c, s, CAT_SMALLCONV, 0, launch_conv_out(c->a_fin_a, c->fin_w, c->fin_bias, f_out, want_grad ? c->a_fin_da : nullptr, M, c->cfg.state_dim, s));
  return 0;
}

// share > 1: eps_out gets ONE row per trajectory, sum_j comb[j] * (input gradient of row j)
int net_backward(ramp_ctx* c, int row0, int R, float* eps_out, hipStream_t s, int share = 1, const float* comb = nullptr) {
  Run r{c, s, R, row0};
  Run rp{c, s, R / share, row0 / share};
  const int nl = c->cfg.n_levels, H = c->cfg.horizon, C0 = c->cfg.unet_input_dim, M = R * H;
  GnBwdArgs g; g.dy = c->a_fin_da; g.x = c->a_fin_c; g.stats = c->a_fin_st; g.gamma = c->fin_g; g.beta = c->fin_b;
  g.dx = c->g_t1; g.R = R; g.L = H; g.C = C0; g.mish = 1;
  float* d = c->g_a; float* e = c->g_b;
  GemmArgs fb = conv5(c->g_t1, C0, c->final_conv.bwd, nullptr, d, C0, M, C0, C0, H, true);
  if (const ramp_ctx::TkcW* w = r.tkc_gn_planes(fb, false)) {
    Run::GnPro p{c->a_fin_c, c->a_fin_st, c->fin_g, c->fin_b};
    fb.A = c->a_fin_da; fb.lda = C0;
    CK(r.tkc(fb, *w, &p, nullptr));
  } else {
  LAUNCH(c, s, CAT_ROW, 0, launch_gn_bwd(g, s));
  CK(r.gemm(fb));
  }
  for (int k = nl - 2; k >= 0; --k) {
    RTB& a = c->rtbs[2 * nl + 2 + 2 * k]; RTB& b = c->rtbs[2 * nl + 3 + 2 * k]; ST& st = c->sts[nl + 1 + k];
    Resample& u = c->ups[k];
    CK(dbg_store(c, "gout/" + u.name, d, (size_t)R * u.Lout * u.C, s));
    {   // dX of Upsample1d: dx[i] = sum_j dy[2i - 1 + j] W_j^T   (4 taps over stride-2 source rows)
      GemmArgs ga = lin(d, u.C, u.w_b, nullptr, e, u.C, R * u.Lin, u.C, u.C);
      ga.taps = 4; ga.shift0 = -1; ga.shift_step = 1; ga.L = u.Lin; ga.a_stride = 2;
      CK(r.gemm(ga)); std::swap(d, e);
    }
    CK(st_backward(r, st, b.a_out, d, e)); std::swap(d, e);
    CK(rtb_backward(r, b, d, e, b.cin, nullptr, 0, nullptr, nullptr)); std::swap(d, e);
    const int lvl = nl - 1 - k;                      // the skip consumed by ups.k
    const int ca = a.cin - c->sts[lvl].C;
    CK(rtb_backward(r, a, d, e, ca, c->skip_grad[lvl], c->sts[lvl].C, nullptr, nullptr)); std::swap(d, e);
  }
  RTB& m1 = c->rtbs[2 * nl]; RTB& m2 = c->rtbs[2 * nl + 1]; ST& ms = c->sts[nl];
  CK(rtb_backward(r, m2, d, e, m2.cin, nullptr, 0, nullptr, nullptr)); std::swap(d, e);
  CK(st_backward(r, ms, m1.a_out, d, e)); std::swap(d, e);
  // the deepest level's transformer output feeds both mid_block1 and (as skip) ups.0
  CK(rtb_backward(r, m1, d, e, m1.cin, nullptr, 0, c->skip_grad[nl - 1], nullptr)); std::swap(d, e);
  for (int k = nl - 1; k >= 0; --k) {
    RTB& a = c->rtbs[2 * k]; RTB& b = c->rtbs[2 * k + 1]; ST& st = c->sts[k];
    if (k < nl - 1) {
      Resample& dn = c->downs[k];
      CK(dbg_store(c, "gout/" + dn.name, d, (size_t)R * dn.Lout * dn.C, s));
      // dX of Downsample1d: dx[2i] = dy[i] W_1 ; dx[2i+1] = dy[i+1] W_0 + dy[i] W_2   (+ skip gradient)
      for (int par = 0; par < 2; ++par) {
        GemmArgs ga = lin(d, dn.C, dn.w_b + (size_t)par * dn.C * dn.C, nullptr, e, dn.C, R * dn.Lout, dn.C, dn.C);
        ga.taps = par ? 2 : 1; ga.shift0 = par; ga.shift_step = -1; ga.L = dn.Lout; ga.c_rstride = 2; ga.c_roff = par;
        if (k >= 1) { ga.resid = c->skip_grad[k]; ga.ldr = dn.C; }
        CK(r.gemm(ga));
      }
      std::swap(d, e);
    }
    const bool pre = share > 1 && k == 0;
    CK(st_backward(r, st, b.a_out, d, e, pre ? share : 1, comb)); std::swap(d, e);
    CK(rtb_backward(pre ? rp : r, b, d, e, b.cin, nullptr, 0, nullptr, nullptr)); std::swap(d, e);
    if (k > 0) { CK(rtb_backward(r, a, d, e, a.cin, nullptr, 0, nullptr, nullptr)); std::swap(d, e); }
    else CK(rtb_backward(pre ? rp : r, a, d, nullptr, 0, nullptr, 0, nullptr, eps_out));
  }
  return 0;
}

// ---- scene encoders (once per scene) ------------------------------------------------------------
GemmArgs slin(const float* A, int K, const float* W, const float* bias, float* C, int M, int N, const float* resid = nullptr) {
  GemmArgs a = lin(A, K, W, bias, C, N, M, N, K);
  if (resid) { a.resid = resid; a.ldr = N; }
  return a;
}
int sw(ramp_ctx* c, const std::string& key, float** out) {
  auto it = c->raw.find("scene_encoder." + key);
  RAMP_REQUIRE(it != c->raw.end(), "missing scene-encoder weight '" + key + "'");
  *out = it->second.first;
  return 0;
}
int scene_scratch(ramp_ctx* c, size_t n) {
  if (n > c->scene_ws_cap) { CK(dev_alloc(c, &c->scene_ws, n)); c->scene_ws_cap = n; }
  return 0;
}

// ObstacleEncoderSet.forward for one scene (obstacle_encoder.py:125-152); cloud (No, Np, 2) -> latent (320)
int encode_scene_2d(ramp_ctx* c, const float* cloud, int No, int Np, float* out, hipStream_t s) {
  const int T = No * Np;
  CK(scene_scratch(c, (size_t)T * (192 + 64 * 4 + 192 + 256) + 4096));
  float* p = c->scene_ws;
  float* F = p; p += (size_t)T * 192;
  float* COMB = p; p += (size_t)T * 64;
  float* X = p; p += (size_t)T * 64;
  float* X2 = p; p += (size_t)T * 64;
  float* LNb = p; p += (size_t)T * 64;
  float* QKV = p; p += (size_t)T * 192;
  float* Hb = p; p += (size_t)T * 256;
  float* centers = p; p += 2 * ((No + 3) & ~3);
  float* maxd = p; p += (No + 3) & ~3;
  float* P0 = p; p += 64; float* P1 = p; p += 256;
  float *w, *b, *g, *be, *div;
  CK(sw(c, "pos_encoder.div_term", &div));
  CK(sw(c, "point_embedding.0.weight", &w)); CK(sw(c, "point_embedding.0.bias", &b));
  CK(sw(c, "point_embedding.1.weight", &g)); CK(sw(c, "point_embedding.1.bias", &be));
  CK(scene_enc2d_prep(cloud, No, Np, centers, maxd, s));
  CK(scene_enc2d_feat(cloud, centers, maxd, div, w, b, g, be, F, Np, T, s));
  CK(sw(c, "combined_encoder.0.weight", &w)); CK(sw(c, "combined_encoder.0.bias", &b));
  CK(launch_gemm(slin(F, 192, w, b, X2, T, 64), s));
  CK(sw(c, "combined_encoder.1.weight", &g)); CK(sw(c, "combined_encoder.1.bias", &be));
  CK(scene_ln64(X2, g, be, COMB, T, 1, s));
  const int dims[3] = {64, 96, 160};
  int off = 0;
  for (int i = 0; i < 3; ++i) {
    const float* xin = COMB;
    for (int j = 0; j < 3; ++j) {
      const std::string t = "set_transformers." + std::to_string(i) + "." + std::to_string(j);
      CK(sw(c, t + ".norm1.weight", &g)); CK(sw(c, t + ".norm1.bias", &be));
      CK(scene_ln64(xin, g, be, LNb, T, 0, s));
      CK(sw(c, t + ".attn.qkv.weight", &w));
      CK(launch_gemm(slin(LNb, 64, w, nullptr, QKV, T, 192), s));
      CK(scene_attention(QKV, LNb, T, 4, 16, 0.25f, s));
      CK(sw(c, t + ".attn.proj.weight", &w)); CK(sw(c, t + ".attn.proj.bias", &b));
      CK(launch_gemm(slin(LNb, 64, w, b, X2, T, 64, xin), s));
      CK(sw(c, t + ".norm2.weight", &g)); CK(sw(c, t + ".norm2.bias", &be));
      CK(scene_ln64(X2, g, be, LNb, T, 0, s));
      CK(sw(c, t + ".mlp.0.weight", &w)); CK(sw(c, t + ".mlp.0.bias", &b));
      CK(launch_gemm(slin(LNb, 64, w, b, Hb, T, 256), s));
      CK(scene_affine_act(Hb, nullptr, nullptr, nullptr, Hb, (long)T * 256, 256, 1, s));
      CK(sw(c, t + ".mlp.3.weight", &w)); CK(sw(c, t + ".mlp.3.bias", &b));
      CK(launch_gemm(slin(Hb, 256, w, b, X, T, 64, X2), s));
      xin = X;
    }
    const int d = dims[i];
    const std::string q = "poolings." + std::to_string(i);
    CK(scene_colreduce(X, P0, 1, T, 64, 0, s));
    CK(sw(c, q + ".0.weight", &w)); CK(sw(c, q + ".0.bias", &b));
    CK(launch_gemm(slin(P0, 64, w, b, P1, 1, d), s));
    CK(scene_affine_act(P1, nullptr, nullptr, nullptr, P1, d, d, 1, s));
    CK(sw(c, q + ".2.weight", &w)); CK(sw(c, q + ".2.bias", &b));
    CK(launch_gemm(slin(P1, d, w, b, out + off, 1, d), s));
    off += d;
  }
  return 0;
}

// ObstacleEncoder.forward, eval mode, one scene (obstacle_encoder3d.py:77-94); cloud (No, Np, 3) -> latent (256)
int encode_scene_3d(ramp_ctx* c, const float* cloud, int No, int Np, float* out, hipStream_t s) {
  const int T = No * Np, E = 256;
  CK(scene_scratch(c, (size_t)T * (64 + 256) + (size_t)No * (256 * 4 + 768 + 512) + 4096));
  float* p = c->scene_ws;
  float* H1 = p; p += (size_t)T * 64;
  float* H2 = p; p += (size_t)T * E;
  float* X = p; p += (size_t)No * E; float* X2 = p; p += (size_t)No * E;
  float* LNb = p; p += (size_t)No * E; float* O = p; p += (size_t)No * E;
  float* QKV = p; p += (size_t)No * 768; float* Hb = p; p += (size_t)No * 512;
  float* sc = p; p += 256; float* sh = p; p += 256; float* S0 = p; p += 256; float* S1 = p; p += 256;
  float *w, *b, *g, *be, *rm, *rv;
  CK(sw(c, "point_processor.conv1.weight", &w)); CK(sw(c, "point_processor.conv1.bias", &b));
  CK(scene_linear_small(cloud, w, b, H1, T, 64, 3, s));
  CK(sw(c, "point_processor.bn1.weight", &g)); CK(sw(c, "point_processor.bn1.bias", &be));
  CK(sw(c, "point_processor.bn1.running_mean", &rm)); CK(sw(c, "point_processor.bn1.running_var", &rv));
  CK(scene_bn_fold(g, be, rm, rv, sc, sh, 64, s));
  CK(scene_affine_act(H1, sc, sh, nullptr, H1, (long)T * 64, 64, 2, s));
  CK(sw(c, "point_processor.conv2.weight", &w)); CK(sw(c, "point_processor.conv2.bias", &b));
  CK(launch_gemm(slin(H1, 64, w, b, H2, T, E), s));
  CK(sw(c, "point_processor.bn2.weight", &g)); CK(sw(c, "point_processor.bn2.bias", &be));
  CK(sw(c, "point_processor.bn2.running_mean", &rm)); CK(sw(c, "point_processor.bn2.running_var", &rv));
  CK(scene_bn_fold(g, be, rm, rv, sc, sh, E, s));
  CK(scene_affine_act(H2, sc, sh, nullptr, H2, (long)T * E, E, 2, s));
  CK(scene_colreduce(H2, X, No, Np, E, 1, s));                     // max over the points of each obstacle
  for (int i = 0; i < 2; ++i) {
    const std::string t = "set_transformer_blocks." + std::to_string(i);
    CK(sw(c, t + ".norm1.weight", &g)); CK(sw(c, t + ".norm1.bias", &be));
    CK(launch_ln_fwd(X, g, be, LNb, No, s));
    CK(sw(c, t + ".mha.in_proj_weight", &w)); CK(sw(c, t + ".mha.in_proj_bias", &b));
    CK(launch_gemm(slin(LNb, E, w, b, QKV, No, 768), s));
    CK(scene_attention(QKV, O, No, 4, 64, 0.125f, s));
    CK(sw(c, t + ".mha.out_proj.weight", &w)); CK(sw(c, t + ".mha.out_proj.bias", &b));
    CK(launch_gemm(slin(O, E, w, b, X2, No, E, X), s));
    CK(sw(c, t + ".norm2.weight", &g)); CK(sw(c, t + ".norm2.bias", &be));
    CK(launch_ln_fwd(X2, g, be, LNb, No, s));
    CK(sw(c, t + ".ffn.0.weight", &w)); CK(sw(c, t + ".ffn.0.bias", &b));
    CK(launch_gemm(slin(LNb, E, w, b, Hb, No, 512), s));
    CK(scene_affine_act(Hb, nullptr, nullptr, nullptr, Hb, (long)No * 512, 512, 2, s));
    CK(sw(c, t + ".ffn.3.weight", &w)); CK(sw(c, t + ".ffn.3.bias", &b));
    CK(launch_gemm(slin(Hb, 512, w, b, X, No, E, X2), s));
  }
  CK(sw(c, "output_proj.weight", &w)); CK(sw(c, "output_proj.bias", &b));
  CK(launch_gemm(slin(X, E, w, b, O, No, E), s));
  CK(scene_colreduce(O, S0, 1, No, E, 1, s));
  CK(sw(c, "global_pooling.0.weight", &w)); CK(sw(c, "global_pooling.0.bias", &b));
  CK(launch_gemm(slin(S0, E, w, b, S1, 1, E), s));
  CK(scene_affine_act(S1, nullptr, nullptr, nullptr, S1, E, E, 2, s));
  CK(sw(c, "global_pooling.2.weight", &w)); CK(sw(c, "global_pooling.2.bias", &b));
  CK(launch_gemm(slin(S1, E, w, b, out, 1, E), s));
  return 0;
}


int ensure_sampler_buffers(ramp_ctx* c, int B, int n_rp, int n_steps, bool chain) {
  const size_t HS = (size_t)c->cfg.horizon * c->cfg.state_dim;
  bool moved = false;
  if ((size_t)B > c->s_cap_B) {
    CK(dev_alloc(c, &c->s_x, B * HS)); CK(dev_alloc(c, &c->s_mean, B * HS)); CK(dev_alloc(c, &c->s_x0, B * HS));
    c->s_cap_B = B; moved = true;
  }
  if ((size_t)B * n_rp > c->s_cap_rows) {
    CK(dev_alloc(c, &c->s_eps, (size_t)B * n_rp * HS)); c->s_cap_rows = (size_t)B * n_rp; moved = true;
  }
  const size_t nn = (size_t)(n_steps + 1) * B * HS;
  if (nn > c->s_noise_cap) { CK(dev_alloc(c, &c->s_noise, nn)); c->s_noise_cap = nn; moved = true; }
  if (chain && nn > c->s_chain_cap) { CK(dev_alloc(c, &c->s_chain, nn)); c->s_chain_cap = nn; moved = true; }
  if (moved) { c->graph_key.clear(); c->r_key.clear(); }      // captured nodes hold the old pointers
  return 0;
}

// one score evaluation over all rows of a batch, chunked to the context capacity
// comb != nullptr (n_rp host floats; sampling jobs): the rows of a trajectory share the network prefix they have in
// common and eps_out receives (B, H, S) = sum_j comb[j] * eps of row j instead of the (B n_rp, H, S) row gradients
int score_all(ramp_ctx* c, const float* x, int B, int n_rp, int t, float* f_out, float* eps_out, hipStream_t s,
              const float* comb = nullptr) {
  const int H = c->cfg.horizon, S = c->cfg.state_dim;
  RAMP_REQUIRE(c->finalized, "weights not finalized");
  RAMP_REQUIRE(c->time_table != nullptr && t >= 0 && t < c->tt_T, "timestep outside the prepared time table");
  RAMP_REQUIRE(c->cross_bias != nullptr, "ramp_set_scene has not been called");
  RAMP_REQUIRE(B > 0 && n_rp >= 1 && n_rp <= 3, "bad batch");
  RAMP_REQUIRE(c->row_variant_cap >= B * n_rp, "row-variant table shorter than the batch (call ramp_set_scene after sizing)");
  const int cap_traj = c->cfg.max_rows / n_rp;
  RAMP_REQUIRE(cap_traj >= 1, "max_rows smaller than n_rp");
  for (int b0 = 0; b0 < B; b0 += cap_traj) {
    const int nb = std::min(cap_traj, B - b0), R = nb * n_rp, row0 = b0 * n_rp;
    float* fo = f_out ? f_out + (size_t)row0 * H * S : nullptr;
    c->site = 0;                                       // every chunk walks the same GEMM call sites
    const int share = (comb && n_rp > 1 && c->share_prefix) ? n_rp : 1;
    RAMP_REQUIRE(!comb || (eps_out && !f_out), "combined evaluation: gradient only");
    CK(net_forward(c, x + (size_t)b0 * H * S, row0, R, n_rp, t, fo, eps_out != nullptr, s, share));
    if (eps_out && share > 1) CK(net_backward(c, row0, R, eps_out + (size_t)b0 * H * S, s, share, comb));
    else if (eps_out) CK(net_backward(c, row0, R, eps_out + (size_t)row0 * H * S, s));
  }
  return 0;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

const char* ramp_last_error(void) { return last_error_cstr(); }
int ramp_version(void) { return 1; }

int ramp_create(const ramp_config* cfg, ramp_ctx** out) {
  RAMP_REQUIRE(cfg && out, "null argument");
  RAMP_REQUIRE(cfg->state_dim >= 2 && cfg->state_dim <= 16, "state_dim out of range");
  RAMP_REQUIRE(cfg->n_levels == 4, "only UNET_DIM_MULTS[1] = (1,2,4,8) is built");
  RAMP_REQUIRE(cfg->unet_input_dim == 32, "unet_input_dim must be 32");
  RAMP_REQUIRE(cfg->horizon >= 8 && cfg->horizon <= 64 && cfg->horizon % 8 == 0, "horizon (n_support_points) must be a multiple of 8 in [8, 64] (three stride-2 levels; attention tiles of at most 64 tokens)");
  RAMP_REQUIRE(cfg->context_dim > 0 && cfg->context_dim <= 512, "context_dim out of range");
  RAMP_REQUIRE(cfg->max_rows >= 1, "max_rows must be positive");
  int ndev = 0;
  RAMP_HIP_CHECK(hipGetDeviceCount(&ndev));
  RAMP_REQUIRE(ndev > 0, "no HIP device: the RAMP sampler has no CPU fallback");
  auto* c = new ramp_ctx();
  c->cfg = *cfg;
  RAMP_HIP_CHECK(hipGetDevice(&c->device));
  {   // the environment only supplies DEFAULTS of the launch plan / arithmetic mode, read here and nowhere below
    const char* env = getenv("RAMP_GEMM_MODE");
    c->gemm_mode = cfg->gemm_mode == 1 ? 0 : cfg->gemm_mode == 2 ? 1 : cfg->gemm_mode == 3 ? 2
                 : (env && std::string(env) == "fp16x3") ? 2 : (env && std::string(env) == "bf16x6") ? 1
                 : (env && std::string(env) == "fp32") ? 0 : RAMP_DEFAULT_GEMM_MODE;
    const char* pe = getenv("RAMP_X6_PIPE");
    c->x6_pipe = !(pe && pe[0] == '0');
    const char* fe = getenv("RAMP_FF_FUSED");
    if (fe) c->ff_fused = atoi(fe);
    const char* xe = getenv("RAMP_FFX");
    if (xe) c->ffx_min_rows = atoi(xe);
    if (const char* me = getenv("RAMP_MFMA16")) c->mfma16 = atoi(me) != 0;
    const char* ke = getenv("RAMP_TKL");
    if (ke) c->tkl_min_rows = atoi(ke);
    const char* tce = getenv("RAMP_TKC");
    if (tce) c->tkc_min_rows = atoi(tce);
    if (const char* tg = getenv("RAMP_TKC_GN")) c->tkc_gn = atoi(tg) & 3;
    const char* twe = getenv("RAMP_TKW");
    if (twe) c->tkw_min_rows = atoi(twe);
    const char* ate = getenv("RAMP_ATK");
    if (ate) c->atk_min_rows = atoi(ate);
    const char* se = getenv("RAMP_SHARE_PREFIX");
    if (se) c->share_prefix = atoi(se) != 0;
    const char* te = getenv("RAMP_X6_THREE");
    if (te) c->three_blocks = te[0] != '0';
    c->prof_dump = getenv("RAMP_PROFILE_DUMP") != nullptr;
    const char* ae = getenv("RAMP_FFX_ABLATE");              // diagnostic A/B of the fused feed-forward inside a whole job
    if (ae) {   // only the variants that compute the SAME result may ride a product launch (16: LayerNorm tables from LDS, 32: no
                // bounding wait at the tile switch, 48: both); the result-changing twins (1, 2, 4, 8, ...) stay in ramp_bench_gemm
      const int av = atoi(ae);
      if (!(av == 0 || av == 16 || av == 32 || av == 48)) {
        delete c;
        RAMP_REQUIRE(false, "RAMP_FFX_ABLATE accepts 0, 16, 32 or 48 (result-preserving variants) on a context; the others change results and exist in ramp_bench_gemm only");
      }
      c->ffx_ablate = av;
    }
    c->tklb_off = getenv("RAMP_TKLB_OFF") != nullptr;
    if (const char* e = getenv("RAMP_ABL")) c->abl_on = atoi(e) != 0;      // diagnostic: 0 = attention backward and d(ln1) as separate launches       // diagnostic: d(ln1) + LN1 backward on the tile kernel + ln_bwd pair
  }
  *out = c;
  return 0;
}

int ramp_get_launch_plan(ramp_ctx* c, ramp_launch_plan* out) {
  RAMP_REQUIRE(c && out, "null argument");
  *out = ramp_launch_plan{c->ff_fused, c->ffx_min_rows, c->share_prefix, c->three_blocks, c->x6_pipe, c->tkl_min_rows, c->atk_min_rows, c->tkc_min_rows, c->tkw_min_rows, c->mfma16};
  return 0;
}
int ramp_set_launch_plan(ramp_ctx* c, const ramp_launch_plan* p) {
  RAMP_REQUIRE(c && p, "null argument");
  RAMP_REQUIRE(p->ff_fused_rows >= 0 && p->ffx_rows >= 0 && p->tkl_rows >= 0 && p->atk_rows >= 0 && p->tkc_rows >= 0 && p->tkw_rows >= 0, "row thresholds must be >= 0");
  RAMP_REQUIRE(!c->finalized || (p->x6_pipe != 0) == (c->x6_pipe != 0), "x6_pipe is fixed once the weights are packed (ramp_finalize_weights)");
  const bool changed = p->ff_fused_rows != c->ff_fused || p->ffx_rows != c->ffx_min_rows || (p->share_prefix != 0) != (c->share_prefix != 0) ||
                       (p->three_blocks != 0) != (c->three_blocks != 0) || p->tkl_rows != c->tkl_min_rows || p->atk_rows != c->atk_min_rows ||
                       p->tkc_rows != c->tkc_min_rows || p->tkw_rows != c->tkw_min_rows || (p->mfma16 != 0) != (c->mfma16 != 0);
  c->ff_fused = p->ff_fused_rows; c->ffx_min_rows = p->ffx_rows; c->tkl_min_rows = p->tkl_rows; c->share_prefix = p->share_prefix != 0;
  c->atk_min_rows = p->atk_rows; c->tkc_min_rows = p->tkc_rows; c->tkw_min_rows = p->tkw_rows;
  c->three_blocks = p->three_blocks != 0; c->x6_pipe = p->x6_pipe != 0; c->mfma16 = p->mfma16 != 0;
  if (changed && c->finalized) {   // other kernels from here on: captured graphs and kept calibrations belong to the old plan
    c->graph_key.clear(); c->r_key.clear();
    c->score_calibrated = false; c->r_calibrated = false; c->s_calibrated = false; c->s_pending = false; c->c_cal_valid = false;
  }
  return 0;
}

int ramp_destroy(ramp_ctx* c) {
  if (!c) return 0;
  for (auto& g : c->graph_exec) if (g) (void)hipGraphExecDestroy(g);
  for (auto& g : c->r_graph) if (g) (void)hipGraphExecDestroy(g);
  if (c->d_ptr_tables) (void)hipFree(c->d_ptr_tables);
  delete c;
  return 0;
}

int ramp_load_weight(ramp_ctx* c, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
  RAMP_REQUIRE(c && name && data && (shape || ndim == 0), "null argument");
  RAMP_REQUIRE(!c->finalized, "weights already finalized");
  std::string key(name);
  size_t n = 1; std::vector<int64_t> shp;
  for (int i = 0; i < ndim; ++i) { RAMP_REQUIRE(shape[i] > 0, "bad shape"); n *= (size_t)shape[i]; shp.push_back(shape[i]); }
  float* d; CK(dev_alloc(c, &d, n));
  RAMP_HIP_CHECK(hipMemcpy(d, data, n * sizeof(float), hipMemcpyHostToDevice));
  c->raw[key] = {d, shp};
  return 0;
}

int ramp_finalize_weights(ramp_ctx* c) {
  RAMP_REQUIRE(c && !c->finalized, "bad context state");
  {
    c->geglu_group = (c->gemm_mode >= 1 && c->x6_pipe) ? 32 : 64;
  }
  const int nl = c->cfg.n_levels, S = c->cfg.state_dim, H = c->cfg.horizon, C0 = c->cfg.unet_input_dim;
  const size_t cap = (size_t)c->cfg.max_rows;
  std::vector<int> dims = {S};
  for (int k = 0; k < nl; ++k) dims.push_back(C0 << k);
  int L = H, tb = 0;
  auto add_rtb = [&](const std::string& name, int ci, int co, int len, bool first) {
    RTB r; r.name = name; r.cin = ci; r.cout = co; r.L = len; r.has_res = ci != co; r.first = first; r.tb_off = tb;
    tb += co; c->rtbs.push_back(r);
  };
  int blk = 0;
  auto add_st = [&](const std::string& name, int ch, int len) {
    ST s; s.name = name; s.C = ch; s.L = len; s.blk0 = blk; blk += 2; c->sts.push_back(s);
  };
  for (int k = 0; k < nl; ++k) {
    const std::string p = "downs." + std::to_string(k);
    add_rtb(p + ".0", dims[k], dims[k + 1], L, k == 0);
    add_rtb(p + ".1", dims[k + 1], dims[k + 1], L, false);
    add_st(p + ".3", dims[k + 1], L);
    if (k < nl - 1) {
      Resample d; d.name = p + ".4"; d.C = dims[k + 1]; d.Lin = L; d.Lout = L / 2; d.taps = 3;
      c->downs.push_back(d); L /= 2;
    }
  }
  RAMP_REQUIRE(dims[1] != S, "first block must have a residual conv");
  const int mid = dims[nl];
  add_rtb("mid_block1", mid, mid, L, false);
  add_rtb("mid_block2", mid, mid, L, false);
  add_st("mid_attention", mid, L);
  for (int k = 0; k < nl - 1; ++k) {
    const std::string p = "ups." + std::to_string(k);
    const int ci = dims[nl - 1 - k], co = dims[nl - k];      // (dim_in, dim_out) of reversed(in_out[1:])
    add_rtb(p + ".0", co * 2, ci, L, false);
    add_rtb(p + ".1", ci, ci, L, false);
    add_st(p + ".3", ci, L);
    Resample u; u.name = p + ".4"; u.C = ci; u.Lin = L; u.Lout = 2 * L; u.taps = 4;
    c->ups.push_back(u); L *= 2;
  }
  RAMP_REQUIRE(L == H, "level bookkeeping");
  c->tt_stride = tb; c->n_blocks_total = blk;
  for (auto& r : c->rtbs) CK(build_rtb(c, r));
  for (auto& s : c->sts) CK(build_st(c, s));
  for (auto& d : c->downs) {
    float *raw, *t;
    CK(get_raw(c, d.name + ".conv.weight", {d.C, d.C, 3}, &raw));
    CK(get_raw(c, d.name + ".conv.bias", {d.C}, &d.bias));
    CK(permute3(c, raw, d.C, d.C, 3, 2, 0, 1, &d.w_f));      // forward taps [j][co][ci] = W[co][ci][j]
    CK(permute3(c, raw, d.C, d.C, 3, 2, 1, 0, &t));          // dX taps      [j][ci][co]
    // even output rows use tap 1, odd rows taps (0, 2): store [W_1 | W_0 | W_2]
    const size_t cc2 = (size_t)d.C * d.C;
    CK(dev_alloc(c, &d.w_b, 3 * cc2));
    RAMP_HIP_CHECK(hipMemcpy(d.w_b, t + cc2, cc2 * 4, hipMemcpyDeviceToDevice));
    RAMP_HIP_CHECK(hipMemcpy(d.w_b + cc2, t, cc2 * 4, hipMemcpyDeviceToDevice));
    RAMP_HIP_CHECK(hipMemcpy(d.w_b + 2 * cc2, t + 2 * cc2, cc2 * 4, hipMemcpyDeviceToDevice));
    CK(dev_alloc(c, &d.a_y, cap * d.Lout * d.C));
  }
  for (auto& u : c->ups) {
    float *raw, *t;
    CK(get_raw(c, u.name + ".conv.weight", {u.C, u.C, 4}, &raw));   // ConvTranspose1d (Cin,Cout,4)
    CK(get_raw(c, u.name + ".conv.bias", {u.C}, &u.bias));
    CK(permute3(c, raw, u.C, u.C, 4, 2, 1, 0, &t));          // forward taps [j][co][ci] = W[ci][co][j]
    // even output rows use taps (1, 3), odd rows taps (0, 2): store [W_1 | W_3 | W_0 | W_2]
    const size_t cc2 = (size_t)u.C * u.C;
    CK(dev_alloc(c, &u.w_f, 4 * cc2));
    const int order[4] = {1, 3, 0, 2};
    for (int q = 0; q < 4; ++q)
      RAMP_HIP_CHECK(hipMemcpy(u.w_f + q * cc2, t + order[q] * cc2, cc2 * 4, hipMemcpyDeviceToDevice));
    CK(permute3(c, raw, u.C, u.C, 4, 2, 0, 1, &u.w_b));      // dX taps [j][ci][co] = W[ci][co][j]
    CK(dev_alloc(c, &u.a_y, cap * u.Lout * u.C));
  }
  CK(pack_conv5(c, "final_conv.0.block.0", C0, C0, &c->final_conv));
  CK(get_raw(c, "final_conv.0.block.2.weight", {C0}, &c->fin_g));
  CK(get_raw(c, "final_conv.0.block.2.bias", {C0}, &c->fin_b));
  CK(get_raw(c, "final_conv.1.weight", {S, C0, 1}, &c->fin_w));
  CK(get_raw(c, "final_conv.1.bias", {S}, &c->fin_bias));
  for (const char* k : {"time_mlp.encoder.1.weight", "time_mlp.encoder.1.bias", "time_mlp.encoder.3.weight",
                        "time_mlp.encoder.3.bias"})
    RAMP_REQUIRE(c->raw.count(k), std::string("missing weight '") + k + "'");
  CK(dev_alloc(c, &c->a_fin_c, cap * H * C0)); CK(dev_alloc(c, &c->a_fin_st, cap * 16));
  CK(dev_alloc(c, &c->a_fin_a, cap * H * C0)); CK(dev_alloc(c, &c->a_fin_da, cap * H * C0));
  // shared temporaries
  size_t max_lc = 0, max_l = 0;
  for (auto& r : c->rtbs) max_lc = std::max(max_lc, (size_t)r.L * std::max(r.cin, r.cout));
  for (auto& s : c->sts) max_l = std::max(max_l, (size_t)s.L);
  max_lc = std::max(max_lc, (size_t)H * C0);
  CK(dev_alloc(c, &c->t_res, cap * max_lc)); CK(dev_alloc(c, &c->t_xn, cap * max_lc));
  CK(dev_alloc(c, &c->t_ln, cap * max_l * 256)); CK(dev_alloc(c, &c->t_o, cap * max_l * 256));
  CK(dev_alloc(c, &c->t_hg, cap * max_l * 1024)); CK(dev_alloc(c, &c->t_dag, cap * max_l * 2048));
  CK(dev_alloc(c, &c->t_dqkv, cap * max_l * 768)); CK(dev_alloc(c, &c->t_dln, cap * max_l * 256));
  CK(dev_alloc(c, &c->t_dz, cap * max_l * 256)); CK(dev_alloc(c, &c->t_dz1, cap * max_l * 256));
  CK(dev_alloc(c, &c->g_a, cap * max_lc)); CK(dev_alloc(c, &c->g_b, cap * max_lc));
  CK(dev_alloc(c, &c->g_t1, cap * max_lc)); CK(dev_alloc(c, &c->g_t2, cap * max_lc)); CK(dev_alloc(c, &c->g_tr, cap * max_lc));
  c->skip_grad.assign(nl, nullptr);
  for (int k = 1; k < nl; ++k) CK(dev_alloc(c, &c->skip_grad[k], cap * c->sts[k].L * c->sts[k].C));
  // bf16x6 planes of every GEMM weight with N >= 128
  if (c->gemm_mode >= 1) {
    auto reg = [&](const float* w, size_t n, int K) -> int {
      if (!w || c->x6.count(w)) return 0;
      float *p, *q = nullptr;
      CK(dev_alloc(c, &p, (3 * n + 1) / 2 + 4));
      CK(launch_split3(w, reinterpret_cast<unsigned short*>(p), (long)n, 0));
      if (n % K == 0 && (n / K) % 32 == 0 && K % 16 == 0) {
        CK(dev_alloc(c, &q, (3 * n + 1) / 2 + 4));
        CK(launch_pack_x6(w, reinterpret_cast<unsigned short*>(q), (long)(n / K), K, 0));
      }
      float* q3 = nullptr; float wsi = 1.f;
      if (q && c->gemm_mode == 2) {
        // static power-of-two weight scale: max |w| -> [2^10, 2^11)
        std::vector<float> hw(n);
        RAMP_HIP_CHECK(hipMemcpy(hw.data(), w, n * sizeof(float), hipMemcpyDeviceToHost));
        float mx = 0.f;
        for (float v : hw) mx = std::max(mx, std::fabs(v));
        float sc = 1.f;
        if (mx > 0.f && std::isfinite(mx)) { int e; std::frexp(mx, &e); sc = std::ldexp(1.f, 11 - e); }   // mx in [2^(e-1), 2^e)
        CK(dev_alloc(c, &q3, n + 4));
        CK(launch_pack_h3(w, reinterpret_cast<unsigned short*>(q3), (long)(n / K), K, sc, 0));
        wsi = 1.f / sc;
      }
      c->x6[w] = {reinterpret_cast<unsigned short*>(p), n, K, reinterpret_cast<unsigned short*>(q),
                  reinterpret_cast<unsigned short*>(q3), wsi};
      return 0;
    };
    if (c->gemm_mode == 2) {
      const size_t nt = (size_t)ramp_ctx::N_OBS_TABLES * ramp_ctx::MAX_SITES;
      float* o; CK(dev_alloc(c, &o, nt + 4));
      c->obs = o; c->range_flag = reinterpret_cast<int*>(o + nt);
      RAMP_HIP_CHECK(hipMemset(o, 0, (nt + 4) * sizeof(float)));
    }
    for (auto& r : c->rtbs) {
      if (!r.first) { CK(reg(r.c1.fwd, 5ul * r.cin * r.cout, r.cin)); CK(reg(r.c1.bwd, 5ul * r.cin * r.cout, r.cout)); }
      CK(reg(r.c2.fwd, 5ul * r.cout * r.cout, r.cout)); CK(reg(r.c2.bwd, 5ul * r.cout * r.cout, r.cout));
      if (r.has_res && !r.first) { CK(reg(r.res_f, (size_t)r.cin * r.cout, r.cin)); CK(reg(r.res_b, (size_t)r.cin * r.cout, r.cout)); }
    }
    for (auto& st : c->sts) {
      CK(reg(st.wpi_f, 256ul * st.C, st.C)); CK(reg(st.wpi_b, 256ul * st.C, 256));
      CK(reg(st.wpo_f, 256ul * st.C, 256)); CK(reg(st.wpo_b, 256ul * st.C, st.C));
      for (auto& k : st.blk) {
        CK(reg(k.wqkv_f, 768ul * 256, 256)); CK(reg(k.wqkv_b, 768ul * 256, 768));
        CK(reg(k.wo_f, 256ul * 256, 256)); CK(reg(k.wo_b, 256ul * 256, 256));
        CK(reg(k.w1_pk, 2048ul * 256, 256)); CK(reg(k.w1_b, 2048ul * 256, 2048));
        CK(reg(k.w2_f, 1024ul * 256, 1024)); CK(reg(k.w2_b, 1024ul * 256, 256));
      }
    }
    if (c->gemm_mode == 2 && c->x6_pipe && c->geglu_group == 32) {
      // weight streams of the token-owning fused feed-forward: first products from the fragment planes the tile kernels use,
      // second products re-packed with the k order of the first one's accumulator rows (ffx_pack_second)
      float* tmp; CK(dev_alloc(c, &tmp, 2048 * 256));
      unsigned short *p2f, *p2b;
      { float* q; CK(dev_alloc(c, &q, 256 * 1024 + 4)); p2f = reinterpret_cast<unsigned short*>(q); }
      { float* q; CK(dev_alloc(c, &q, 256 * 2048 + 4)); p2b = reinterpret_cast<unsigned short*>(q); }
      unsigned short* p1x; { float* q; CK(dev_alloc(c, &q, 256 * 2048 + 4)); p1x = reinterpret_cast<unsigned short*>(q); }
      for (auto& st : c->sts)
        for (auto& k : st.blk) {
          const auto& e1 = c->x6.at(k.w1_pk); const auto& e1b = c->x6.at(k.w1_b);
          const auto& e2 = c->x6.at(k.w2_f); const auto& e2b = c->x6.at(k.w2_b);
          RAMP_REQUIRE(e1.packed3 && e2b.packed3 && e1.w_scale_inv == e1b.w_scale_inv && e2.w_scale_inv == e2b.w_scale_inv, "ffx: weight planes missing");
          float *sf, *sb; CK(dev_alloc(c, &sf, 96 * 8192 + 4)); CK(dev_alloc(c, &sb, 96 * 8192 + 4));
          k.ffx_f = reinterpret_cast<unsigned short*>(sf); k.ffx_b = reinterpret_cast<unsigned short*>(sb);
          CK(ffx_pack_second(k.w2_f, 256, 1024, 0, 1.f / e2.w_scale_inv, tmp, p2f, 0));
          CK(ffx_build_stream(e1.packed3, p2f, k.ffx_f, false, 0));
          CK(ffx_pack_second(k.w1_b, 256, 2048, 1, 1.f / e1.w_scale_inv, tmp, p2b, 0));
          CK(ffx_build_stream(e2b.packed3, p2b, k.ffx_b, true, 0));
          k.ffx_wsi_w1 = e1.w_scale_inv; k.ffx_wsi_w2 = e2.w_scale_inv;
          // the same two streams in 16 x 32 fragments for the v_mfma_f32_16x16x32_f16 kernels (ffx16.hip), same scales
          float *sf6, *sb6; CK(dev_alloc(c, &sf6, 96 * 8192 + 4)); CK(dev_alloc(c, &sb6, 96 * 8192 + 4));
          k.ffx16_f = reinterpret_cast<unsigned short*>(sf6); k.ffx16_b = reinterpret_cast<unsigned short*>(sb6);
          CK(ffx16_pack(k.w1_pk, 2048, 256, 0, 1.f / e1.w_scale_inv, tmp, p1x, 0));
          CK(ffx16_pack(k.w2_f, 256, 1024, 1, 1.f / e2.w_scale_inv, tmp, p2f, 0));
          CK(ffx16_build_stream(p1x, p2f, k.ffx16_f, false, 0));
          CK(ffx16_pack(k.w2_b, 1024, 256, 0, 1.f / e2.w_scale_inv, tmp, p1x, 0));
          CK(ffx16_pack(k.w1_b, 256, 2048, 2, 1.f / e1.w_scale_inv, tmp, p2b, 0));
          CK(ffx16_build_stream(p1x, p2b, k.ffx16_b, true, 0));
          // ... and the K = 256 attention linears the token-owning kernel serves (LN1 -> QKV, out-projection, d(o)) for tkl16.hip
          for (const auto& wn : {std::make_pair((const float*)k.wqkv_f, 768), std::make_pair((const float*)k.wo_f, 256), std::make_pair((const float*)k.wo_b, 256)}) {
            const auto& ew = c->x6.at(wn.first);
            if (!ew.packed3) continue;
            float* q16; CK(dev_alloc(c, &q16, (size_t)wn.second * 256 + 4));          // 2 planes x N x 256 halves
            CK(ffx16_pack(wn.first, wn.second, 256, 0, 1.f / ew.w_scale_inv, tmp, reinterpret_cast<unsigned short*>(q16), 0));
            c->tk16_w[wn.first] = reinterpret_cast<unsigned short*>(q16);
          }
        }
    }
    if (c->gemm_mode == 2 && c->x6_pipe) {
      // weight streams of the fused self-attention + output projection (atk.hip): Wo in head order, k permuted to the row order
      // of the attention output's accumulator tiles, with the scale the tile kernels' planes of the same weight carry
      for (auto& st : c->sts)
        for (auto& k : st.blk) {
          const auto& eo = c->x6.at(k.wo_f);
          float* q; CK(dev_alloc(c, &q, 8 * 8192 + 4));
          k.ato_w = reinterpret_cast<unsigned short*>(q); k.ato_wsi = eo.w_scale_inv;
          CK(ato_pack(k.wo_f, 1.f / eo.w_scale_inv, k.ato_w, 0));
          const auto& eq = c->x6.at(k.wqkv_b);             // the same for the backward half (atl.hip): Wqkv^T in the order its gradient tiles appear
          float* q2; CK(dev_alloc(c, &q2, 256 * 768 + 4));
          k.abl_w = reinterpret_cast<unsigned short*>(q2);
          CK(abl_pack(k.wqkv_b, 1.f / eq.w_scale_inv, k.abl_w, 0));
        }
    }
    if (c->gemm_mode == 2 && c->x6_pipe) {
      // fragment planes of the narrow k = 5 convolutions (tkc.hip), forward and input-gradient weights: static power-of-two scale,
      // max |w| -> [2^10, 2^11) like every other fp16x3 weight
      auto reg_tkc = [&](const float* w, int N, int K) -> int {
        if (!w || c->tkc_w.count(w) || !((N == 32 || N == 64) && (K == 32 || K == 64))) return 0;
        const size_t n = 5ul * N * K;
        std::vector<float> hw(n);
        RAMP_HIP_CHECK(hipMemcpy(hw.data(), w, n * sizeof(float), hipMemcpyDeviceToHost));
        float mx = 0.f;
        for (float v : hw) mx = std::max(mx, std::fabs(v));
        float sc = 1.f;
        if (mx > 0.f && std::isfinite(mx)) { int e; std::frexp(mx, &e); sc = std::ldexp(1.f, 11 - e); }
        float* q; CK(dev_alloc(c, &q, tkc_packed_halves(N, K) / 2 + 4));
        CK(tkc_pack(w, N, K, sc, reinterpret_cast<unsigned short*>(q), 0));
        c->tkc_w[w] = {reinterpret_cast<unsigned short*>(q), 1.f / sc};
        return 0;
      };
      for (auto& r : c->rtbs) {
        if (!r.first) { CK(reg_tkc(r.c1.fwd, r.cout, r.cin)); CK(reg_tkc(r.c1.bwd, r.cin, r.cout)); }
        CK(reg_tkc(r.c2.fwd, r.cout, r.cout)); CK(reg_tkc(r.c2.bwd, r.cout, r.cout));
      }
      CK(reg_tkc(c->final_conv.fwd, C0, C0)); CK(reg_tkc(c->final_conv.bwd, C0, C0));
    }
    for (auto& d : c->downs) { CK(reg(d.w_f, 3ul * d.C * d.C, d.C)); CK(reg(d.w_b, 3ul * d.C * d.C, d.C)); }
    for (auto& u : c->ups) { CK(reg(u.w_f, 4ul * u.C * u.C, u.C)); CK(reg(u.w_b, 4ul * u.C * u.C, u.C)); }
  }
  RAMP_HIP_CHECK(hipDeviceSynchronize());
  CK(init_gemm_attributes());          // hipFuncSetAttribute calls must not happen inside a graph capture
  CK(init_attention_attributes());
  CK(init_ffx_attributes()); CK(init_ffx16_attributes()); CK(init_tkl16_attributes());
  CK(init_tkl_attributes());
  CK(init_atk_attributes());
  CK(init_atl_attributes());
  CK(init_tkc_attributes());
  CK(init_tkw_attributes());
  c->finalized = true;
  return 0;
}

int ramp_prepare_time_table(ramp_ctx* c, int32_t T, void* stream) {
  RAMP_REQUIRE(c && c->finalized && T > 0 && T <= 4096, "bad arguments");
  hipStream_t s = as_stream(stream);
  const int n = (int)c->rtbs.size();
  std::vector<const float*> cw(n), cb(n); std::vector<int> couts(n), offs(n);
  for (int i = 0; i < n; ++i) {
    float *w, *b;
    CK(get_raw(c, c->rtbs[i].name + ".cond_mlp.1.weight", {c->rtbs[i].cout, 32}, &w));
    CK(get_raw(c, c->rtbs[i].name + ".cond_mlp.1.bias", {c->rtbs[i].cout}, &b));
    cw[i] = w; cb[i] = b; couts[i] = c->rtbs[i].cout; offs[i] = c->rtbs[i].tb_off;
  }
  const size_t bytes = n * (2 * sizeof(void*) + 2 * sizeof(int));
  char* d = nullptr;
  RAMP_HIP_CHECK(hipMalloc(&d, bytes));
  std::vector<char> h(bytes);
  std::memcpy(h.data(), cw.data(), n * sizeof(void*));
  std::memcpy(h.data() + n * sizeof(void*), cb.data(), n * sizeof(void*));
  std::memcpy(h.data() + 2 * n * sizeof(void*), couts.data(), n * sizeof(int));
  std::memcpy(h.data() + 2 * n * sizeof(void*) + n * sizeof(int), offs.data(), n * sizeof(int));
  RAMP_HIP_CHECK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
  if (T > c->tt_T) { CK(dev_alloc(c, &c->time_table, (size_t)T * c->tt_stride)); CK(dev_alloc(c, &c->time_temb, (size_t)T * 32)); }
  TimeTableArgs a;
  a.w1 = c->raw["time_mlp.encoder.1.weight"].first; a.b1 = c->raw["time_mlp.encoder.1.bias"].first;
  a.w2 = c->raw["time_mlp.encoder.3.weight"].first; a.b2 = c->raw["time_mlp.encoder.3.bias"].first;
  a.cond_w = reinterpret_cast<const float* const*>(d);
  a.cond_b = reinterpret_cast<const float* const*>(d + n * sizeof(void*));
  a.couts = reinterpret_cast<const int*>(d + 2 * n * sizeof(void*));
  a.offs = reinterpret_cast<const int*>(d + 2 * n * sizeof(void*) + n * sizeof(int));
  a.n_rtb = n; a.table = c->time_table; a.stride = c->tt_stride; a.T = T; a.temb = c->time_temb;
  int rc = launch_time_table(a, s);
  RAMP_HIP_CHECK(hipStreamSynchronize(s));
  RAMP_HIP_CHECK(hipFree(d));
  if (rc) return rc;
  c->tt_T = T;
  return 0;
}

int ramp_time_embedding(ramp_ctx* c, int32_t t, float* out32, void* stream) {
  RAMP_REQUIRE(c && out32 && c->time_temb && t >= 0 && t < c->tt_T, "time table not prepared for this t (ramp_prepare_time_table)");
  RAMP_HIP_CHECK(hipMemcpyAsync(out32, c->time_temb + (size_t)t * 32, 32 * sizeof(float), hipMemcpyDeviceToDevice, as_stream(stream)));
  return 0;
}

int ramp_set_scene(ramp_ctx* c, const float* latents, int32_t n_variants, const int32_t* row_variant_host,
                   int32_t n_rows_pattern, void* stream) {
  RAMP_REQUIRE(c && c->finalized && latents && n_variants >= 1 && n_variants <= 8, "bad arguments");
  RAMP_REQUIRE(row_variant_host && n_rows_pattern >= 1 && n_rows_pattern <= 64, "bad row-variant pattern");
  for (int i = 0; i < n_rows_pattern; ++i)
    RAMP_REQUIRE(row_variant_host[i] >= 0 && row_variant_host[i] < n_variants, "row variant out of range");
  hipStream_t s = as_stream(stream);
  const int nb = c->n_blocks_total;
  if (n_variants > c->n_variants) { CK(dev_alloc(c, &c->cross_bias, (size_t)n_variants * nb * 256)); c->n_variants = n_variants; }
  std::vector<const float*> wv(nb), wo(nb), bo(nb);
  for (auto& st : c->sts) for (int b = 0; b < 2; ++b) {
    wv[st.blk0 + b] = st.blk[b].wv2; wo[st.blk0 + b] = st.blk[b].wo2; bo[st.blk0 + b] = st.blk[b].bo2;
  }
  char* d = nullptr;
  RAMP_HIP_CHECK(hipMalloc(&d, 3 * nb * sizeof(void*) + n_rows_pattern * sizeof(int)));
  RAMP_HIP_CHECK(hipMemcpy(d, wv.data(), nb * sizeof(void*), hipMemcpyHostToDevice));
  RAMP_HIP_CHECK(hipMemcpy(d + nb * sizeof(void*), wo.data(), nb * sizeof(void*), hipMemcpyHostToDevice));
  RAMP_HIP_CHECK(hipMemcpy(d + 2 * nb * sizeof(void*), bo.data(), nb * sizeof(void*), hipMemcpyHostToDevice));
  RAMP_HIP_CHECK(hipMemcpy(d + 3 * nb * sizeof(void*), row_variant_host, n_rows_pattern * sizeof(int), hipMemcpyHostToDevice));
  int rc = launch_cross_bias(latents, n_variants, c->cfg.context_dim, reinterpret_cast<const float* const*>(d),
                             reinterpret_cast<const float* const*>(d + nb * sizeof(void*)),
                             reinterpret_cast<const float* const*>(d + 2 * nb * sizeof(void*)), nb, c->cross_bias, s);
  // row -> variant table for the largest batch this context may see in one ramp_sample call
  const int want = std::max(c->row_variant_cap, 1 << 20);
  if (rc == 0 && c->row_variant_cap < want) {
    float* p; rc = dev_alloc(c, &p, want); c->row_variant = reinterpret_cast<int*>(p); c->row_variant_cap = want;
  }
  if (rc == 0) {
    hipLaunchKernelGGL(fill_pattern_kernel, dim3(1024), dim3(256), 0, s, c->row_variant,
                       reinterpret_cast<const int*>(d + 3 * nb * sizeof(void*)), n_rows_pattern, c->row_variant_cap);
  }
  RAMP_HIP_CHECK(hipStreamSynchronize(s));
  RAMP_HIP_CHECK(hipFree(d));
  c->graph_key.clear();     // scene changed: cross_bias pointer may have moved
  c->r_key.clear();
  c->score_calibrated = false; c->r_calibrated = false; c->s_calibrated = false; c->s_pending = false; c->c_cal_valid = false;
  return rc;
}

int ramp_encode_scene(ramp_ctx* c, const float* cloud, int32_t n_obstacles, int32_t n_points, int32_t point_dim,
                      float* latent_out, void* stream) {
  RAMP_REQUIRE(c && c->finalized && cloud && latent_out && n_obstacles > 0 && n_points > 0, "bad arguments");
  RAMP_REQUIRE((reinterpret_cast<uintptr_t>(latent_out) & 15) == 0, "latent_out must be 16-byte aligned");
  if (point_dim == 2) {
    RAMP_REQUIRE(c->cfg.context_dim == 320, "2-D scene encoder produces a 320-d latent");
    return encode_scene_2d(c, cloud, n_obstacles, n_points, latent_out, as_stream(stream));
  }
  RAMP_REQUIRE(point_dim == 3 && c->cfg.context_dim == 256, "3-D scene encoder takes (No,Np,3) and produces 256-d");
  return encode_scene_3d(c, cloud, n_obstacles, n_points, latent_out, as_stream(stream));
}

int ramp_score(ramp_ctx* c, const float* x, int32_t B, int32_t n_rp, int32_t t, float* f_out, float* eps_out,
               void* stream) {
  RAMP_REQUIRE(c && x, "null argument");
  c->launches = 0;
  hipStream_t s = as_stream(stream);
  c->s_calibrated = false; c->s_pending = false;          // single evaluations overwrite the tables a kept job calibration reads
  if (c->gemm_mode != 2 || c->force_x6) { c->score_last_mode = c->gemm_mode == 2 ? 1 : c->gemm_mode; return score_all(c, x, B, n_rp, t, f_out, eps_out, s); }
  // fp16x3: the same delayed scaling as inside ramp_sample, with the calibration kept across calls.  The first
  // evaluation after context creation / a scene change / a ramp_sample runs the bf16x6 kernels and records every call
  // site's operand maximum; every later one runs fp16x3 scaled from its predecessor's maxima, and is repeated at once
  // with the bf16x6 kernels (recording again) if the range guard fired -- the caller never sees a flagged result.
  auto tables = [&](int parity) {
    c->obs_out = c->obs + parity * ramp_ctx::MAX_SITES;
    c->obs_in = c->obs + (parity ^ 1) * ramp_ctx::MAX_SITES;
    hipLaunchKernelGGL(zero_words_kernel, dim3(ramp_ctx::MAX_SITES / 256), dim3(256), 0, s,
                       reinterpret_cast<unsigned*>(c->obs_out), ramp_ctx::MAX_SITES);
  };
  int rc = 0;
  // (a forward-only evaluation leaves the call sites of the input-gradient pass without maxima)
  if (c->score_calibrated && (c->score_calibrated_bwd || eps_out == nullptr)) {
    hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(256), 0, s, reinterpret_cast<unsigned*>(c->range_flag), 1);
    tables(c->score_parity);
    RAMP_HIP_CHECK(hipGetLastError());
    c->phase = 2;
    rc = score_all(c, x, B, n_rp, t, f_out, eps_out, s);
    c->phase = 0;
    CK(rc);
    int flag = 0;
    RAMP_HIP_CHECK(hipMemcpyAsync(&flag, c->range_flag, sizeof(int), hipMemcpyDeviceToHost, s));
    RAMP_HIP_CHECK(hipStreamSynchronize(s));
    c->r_calibrated = false;
    if (!flag) { c->score_parity ^= 1; c->score_last_mode = 2; c->score_calibrated_bwd = eps_out != nullptr; return 0; }
  }
  tables(c->score_parity);
  RAMP_HIP_CHECK(hipGetLastError());
  c->phase = 1;
  rc = score_all(c, x, B, n_rp, t, f_out, eps_out, s);
  c->phase = 0;
  CK(rc);
  c->score_calibrated = true; c->score_calibrated_bwd = eps_out != nullptr; c->score_parity ^= 1; c->score_last_mode = 1;
  c->r_calibrated = false;
  return 0;
}

static int sample_body(ramp_ctx* c, const ramp_sample_params* p, hipStream_t s, bool chain, bool steady, int cal_eval = -1) {
  const int B = p->B, H = c->cfg.horizon, S = c->cfg.state_dim;
  const size_t HS = (size_t)H * S, n = (size_t)B * HS;
  HardConds hc; hc.idx = c->s_hard_idx; hc.val = c->s_hard_val; hc.n = p->n_hard;
  // noise_mode 1: the job's whole noise block is drawn here, inside the (captured) job, from the device record {seed, offset}
  if (p->noise_mode == 1)
    LAUNCH(c, s, CAT_SAMPLER, 0, launch_philox_normal_sharded(c->s_noise, p->ddim ? 1 : p->n_steps + 1, B, (int)HS, (long)p->philox_sample0,
                                                              p->philox_total > 0 ? (long)p->philox_total : (long)B, c->s_philox, s));
  // x_T = noise[0]; apply_hard_conditioning; chain[0]
  RAMP_HIP_CHECK(hipMemcpyAsync(c->s_x, c->s_noise, n * 4, hipMemcpyDeviceToDevice, s));
  LAUNCH(c, s, CAT_SAMPLER, 0, launch_hard_cond(c->s_x, hc, B, H, S, s));
  if (chain) RAMP_HIP_CHECK(hipMemcpyAsync(c->s_chain, c->s_x, n * 4, hipMemcpyDeviceToDevice, s));
  ApfArgs ap; ap.cloud = c->s_cloud; ap.window = c->s_window; ap.B = B; ap.H = H; ap.S = S; ap.P = p->apf.n_points;
  ap.win = p->apf.window; ap.thr = p->apf.threshold; ap.strength = p->apf.strength;
  if (c->gemm_mode == 2) {
    hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(256), 0, s, reinterpret_cast<unsigned*>(c->range_flag), 1);
    if (!c->force_x6) hipLaunchKernelGGL(zero_words_kernel, dim3((p->n_steps + 255) / 256), dim3(256), 0, s, reinterpret_cast<unsigned*>(c->trip_log), p->n_steps);
    RAMP_HIP_CHECK(hipGetLastError());
  }
  for (int j = 0; j < p->n_steps; ++j) {
    if (c->gemm_mode == 2 && !c->force_x6) {
      // evaluation 0 runs fp16x3 scaled from the context's CANONICAL maxima (table 2: canonical_calibration below) -- or, with
      // ramp_set_calibration_reuse(ctx, 0), calibrates itself (bf16x6 + recorded operand maxima); evaluation j >= 1 runs fp16x3
      // scaled from j - 1
      // (j == cal_eval, cal_eval + 1: the evaluation the range guard flagged first in this job's previous run calibrates instead -- see
      // ramp_ctx::trip_log -- and so does its successor: an excursion that is gone one step later (a spike) would otherwise trip the guard
      // again from the other side, its successor being scaled from the spike's maxima; 3 % of a job per calibrating evaluation)
      c->phase = ((j == 0 && !steady) || j == cal_eval || (cal_eval >= 0 && j == cal_eval + 1)) ? 1 : 2;
      c->obs_out = c->obs + (j & 1) * ramp_ctx::MAX_SITES;
      c->obs_in = c->obs + (j == 0 ? 2 : ((j & 1) ^ 1)) * ramp_ctx::MAX_SITES;
      hipLaunchKernelGGL(zero_words_kernel, dim3(ramp_ctx::MAX_SITES / 256), dim3(256), 0, s,
                         reinterpret_cast<unsigned*>(c->obs_out), ramp_ctx::MAX_SITES);
      RAMP_HIP_CHECK(hipGetLastError());
    }
    // the weights of the rows' gradients in e_comb (diffusion_model_static.py:164-165, 214): the network prefix the rows
    // of a trajectory have in common is evaluated once and its input gradient once, on the combined gradient
    const float comb[3] = {p->n_rp == 2 ? (float)(1.0 + p->w0) : (float)p->w0, p->n_rp == 2 ? -(float)p->w0 : (float)p->w1,
                           (float)(1.0 - p->w0 - p->w1)};
    const bool shared = p->n_rp > 1 && c->share_prefix;
    const int rc_score = score_all(c, c->s_x, B, p->n_rp, p->t[j], nullptr, c->s_eps, s, shared ? comb : nullptr);
    c->phase = 0;
    CK(rc_score);
    if (c->gemm_mode == 2 && !c->force_x6) {
      hipLaunchKernelGGL(log_flag_kernel, dim3(1), dim3(64), 0, s, c->range_flag, c->trip_log, j);
      RAMP_HIP_CHECK(hipGetLastError());
    }
    CfgMeanArgs m; m.x = c->s_x; m.eps = c->s_eps; m.B = B; m.HS = (int)HS; m.n_rp = shared ? 1 : p->n_rp;
    m.w0 = (float)p->w0; m.w1 = (float)p->w1; m.w0p1 = (float)(1.0 + p->w0);
    m.sqrt_recip = p->sqrt_recip[j]; m.sqrt_recipm1 = p->sqrt_recipm1[j]; m.clip = p->clip_denoised; m.predict_x0 = p->predict_x0 != 0;
    float* chain_j = chain ? c->s_chain + (size_t)(j + 1) * n : nullptr;
    const bool apf = p->apf.cloud != nullptr && p->apply_apf && p->apply_apf[j];
    if (!p->ddim) {
      m.coef1 = p->coef1[j]; m.coef2 = p->coef2[j]; m.mean = c->s_mean; m.x0 = nullptr;
      LAUNCH(c, s, CAT_SAMPLER, 0, launch_cfg_mean(m, s));
      if (apf) { ap.traj = c->s_mean; for (int q = 0; q < std::max(1, p->apf.passes); ++q) LAUNCH(c, s, CAT_SAMPLER, 0, launch_apf(ap, s)); }
      LAUNCH(c, s, CAT_SAMPLER, 0, launch_ddpm_finish(c->s_mean, c->s_noise + (size_t)(j + 1) * n, p->stdv[j], p->noise_scale ? p->noise_scale[j] : 1.f, p->use_noise[j],
                            hc, c->s_x, chain_j, B, H, S, s));
    } else {
      m.mean = nullptr; m.x0 = c->s_x0;
      LAUNCH(c, s, CAT_SAMPLER, 0, launch_cfg_mean(m, s));
      if (apf) {
        ap.traj = c->s_x0;
        for (int q = 0; q < std::max(1, p->apf.passes); ++q) { LAUNCH(c, s, CAT_SAMPLER, 0, launch_apf(ap, s)); LAUNCH(c, s, CAT_SAMPLER, 0, launch_hard_cond(c->s_x0, hc, B, H, S, s)); }
      }
      LAUNCH(c, s, CAT_SAMPLER, 0, launch_ddim_finish(c->s_x, c->s_x0, p->sqrt_a_t[j], p->sqrt_1m_a_t[j], p->sqrt_a_prev[j], p->dir_coef[j], hc,
                            c->s_x, chain_j, B, H, S, s));
    }
  }
  return 0;
}

// The calibration every job's first evaluation is scaled from (fp16x3 mode): ONE score evaluation on the bf16x6 kernels that only records the
// operand maxima of every GEMM call site (phase 1) into table 2, on a CANONICAL input -- x ~ N(0, I) from Philox4x32-10 with a fixed seed, the
// at the job's first timestep -- not on anybody's data, not even the job's start / goal values: x_T of every job is a draw of the same distribution, the
// recorded maxima only pick power-of-two scales (the operand may then grow 2^9.9-fold before the range guard fires, elements down to 2^-8 of the
// maximum keep all 22 bits), and the guard covers a caller whose x_T is something else.  Runs eagerly on the job's stream, outside its graph,
// once per (batch, first timestep, hard-condition layout, scene).
static int canonical_calibration(ramp_ctx* c, const ramp_sample_params* p, hipStream_t s) {
  const int B = p->B, H = c->cfg.horizon, S = c->cfg.state_dim;
  if (!c->c_cal_rec) {
    float* q; CK(dev_alloc(c, &q, 4)); c->c_cal_rec = reinterpret_cast<unsigned long long*>(q);
    const unsigned long long rec[2] = {0x52414d5043414cull /* "RAMPCAL" */, 0ull};
    RAMP_HIP_CHECK(hipMemcpyAsync(c->c_cal_rec, rec, 16, hipMemcpyHostToDevice, s));
    RAMP_HIP_CHECK(hipStreamSynchronize(s));       // (rec is a stack array)
  }
  RAMP_REQUIRE((H * S) % 4 == 0, "H * S must be a multiple of 4");
  // (round 6, ADVICE r5: NO hard conditions on the canonical input -- with the caller's start / goal applied, table 2 depended on the values of
  // whichever job triggered it, and a later job with other values inherited scales a fresh context would not have chosen)
  CK(launch_philox_normal(c->s_x, (long)B * H * S, c->c_cal_rec, s));
  c->phase = 1;
  c->obs_out = c->obs + 2 * ramp_ctx::MAX_SITES; c->obs_in = c->obs;
  hipLaunchKernelGGL(zero_words_kernel, dim3(ramp_ctx::MAX_SITES / 256), dim3(256), 0, s, reinterpret_cast<unsigned*>(c->obs_out), ramp_ctx::MAX_SITES);
  const float comb[3] = {p->n_rp == 2 ? (float)(1.0 + p->w0) : (float)p->w0, p->n_rp == 2 ? -(float)p->w0 : (float)p->w1, (float)(1.0 - p->w0 - p->w1)};
  const bool shared = p->n_rp > 1 && c->share_prefix;
  const int rc = score_all(c, c->s_x, B, p->n_rp, p->t[0], nullptr, c->s_eps, s, shared ? comb : nullptr);
  c->phase = 0;
  return rc;
}

int ramp_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
  RAMP_REQUIRE(out && n > 0, "null argument");
  hipStream_t s = as_stream(stream);
  unsigned long long* rec = nullptr;
  RAMP_HIP_CHECK(hipMalloc(&rec, 16));
  const unsigned long long h[2] = {seed, offset};
  hipError_t e = hipMemcpyAsync(rec, h, 16, hipMemcpyHostToDevice, s);
  int rc = e == hipSuccess ? launch_philox_normal(out, (long)n, rec, s) : -1;
  if (e != hipSuccess) set_last_error(std::string("hipMemcpyAsync failed: ") + hipGetErrorString(e));
  (void)hipStreamSynchronize(s);
  (void)hipFree(rec);
  return rc;
}

int ramp_sample(ramp_ctx* c, const ramp_sample_params* p, const float* noise, float* chain_out, float* x_out,
                void* stream) {
  RAMP_REQUIRE(c && p, "null argument");
  RAMP_REQUIRE(p->noise_mode == 0 || p->noise_mode == 1, "noise_mode must be 0 (injected) or 1 (Philox inside the job)");
  RAMP_REQUIRE(p->philox_total == 0 || (p->philox_sample0 >= 0 && p->philox_sample0 + p->B <= p->philox_total), "philox shard outside the job (philox_sample0 + B <= philox_total)");
  RAMP_REQUIRE(p->noise_mode == 0 || (c->cfg.horizon * c->cfg.state_dim) % 4 == 0, "noise_mode 1 needs H * S to be a multiple of 4");
  RAMP_REQUIRE(noise || p->noise_mode == 1, "null noise (only a job that draws its own, noise_mode 1, may omit it)");
  RAMP_REQUIRE(p->B > 0 && p->n_steps > 0 && p->n_rp >= 1 && p->n_rp <= 3, "bad sample dims");
  RAMP_REQUIRE(p->t && p->sqrt_recip && p->sqrt_recipm1, "missing schedule arrays");
  if (p->ddim) RAMP_REQUIRE(p->sqrt_a_t && p->sqrt_1m_a_t && p->sqrt_a_prev && p->dir_coef, "missing DDIM arrays");
  else RAMP_REQUIRE(p->coef1 && p->coef2 && p->stdv && p->use_noise, "missing DDPM arrays");
  RAMP_REQUIRE(p->n_hard == 0 || (p->hard_idx_host && p->hard_val), "missing hard conditions");
  hipStream_t s = as_stream(stream);
  const int B = p->B, H = c->cfg.horizon, S = c->cfg.state_dim;
  const size_t n = (size_t)B * H * S;
  const bool chain = chain_out != nullptr;
  const size_t n_noise = (p->ddim ? 1 : (size_t)p->n_steps + 1) * n;
  CK(ensure_sampler_buffers(c, B, p->n_rp, p->n_steps, chain));
  if (p->n_steps > c->trip_log_cap) {
    float* q; const int cap = std::max(256, p->n_steps);
    CK(dev_alloc(c, &q, cap)); c->trip_log = reinterpret_cast<int*>(q); c->trip_log_cap = cap; c->graph_key.clear();
  }
  for (int j = 0; j < p->n_hard; ++j) RAMP_REQUIRE(p->hard_idx_host[j] >= 0 && p->hard_idx_host[j] < H, "hard index out of range");
  if (!c->s_hard_idx) { float* q; CK(dev_alloc(c, &q, 256)); c->s_hard_idx = reinterpret_cast<int*>(q); }
  RAMP_REQUIRE(p->n_hard <= 256, "too many hard conditions");
  const size_t hv = (size_t)p->n_hard * B * S;
  if (hv > c->s_hard_val_cap) { CK(dev_alloc(c, &c->s_hard_val, hv)); c->s_hard_val_cap = hv; c->graph_key.clear(); }
  if (p->apf.cloud) {
    RAMP_REQUIRE(p->apf.n_points > 0 && p->apf.window >= 0 && p->apf.window <= 64 && p->apf.window_weights_host, "bad APF params");
    if (!c->s_window) CK(dev_alloc(c, &c->s_window, 256));
    if ((size_t)p->apf.n_points * 2 > c->s_cloud_cap) {
      CK(dev_alloc(c, &c->s_cloud, (size_t)p->apf.n_points * 2)); c->s_cloud_cap = (size_t)p->apf.n_points * 2; c->graph_key.clear();
    }
    RAMP_HIP_CHECK(hipMemcpyAsync(c->s_window, p->apf.window_weights_host, (2 * p->apf.window + 1) * 4, hipMemcpyHostToDevice, s));
    RAMP_HIP_CHECK(hipMemcpyAsync(c->s_cloud, p->apf.cloud, (size_t)p->apf.n_points * 8, hipMemcpyDeviceToDevice, s));
  }
  if (p->n_hard) {
    RAMP_HIP_CHECK(hipMemcpyAsync(c->s_hard_idx, p->hard_idx_host, p->n_hard * 4, hipMemcpyHostToDevice, s));
    RAMP_HIP_CHECK(hipMemcpyAsync(c->s_hard_val, p->hard_val, hv * 4, hipMemcpyDeviceToDevice, s));
  }
  if (p->noise_mode == 1) {
    if (!c->s_philox) { float* q; CK(dev_alloc(c, &q, 4)); c->s_philox = reinterpret_cast<unsigned long long*>(q); }
    const unsigned long long rec[2] = {p->philox_seed, p->philox_offset};
    RAMP_HIP_CHECK(hipMemcpyAsync(c->s_philox, rec, 16, hipMemcpyHostToDevice, s));     // (pageable host memory: the copy is staged before the call returns)
  } else {
    RAMP_HIP_CHECK(hipMemcpyAsync(c->s_noise, noise, n_noise * 4, hipMemcpyDeviceToDevice, s));
  }
  c->launches = 0;
  c->score_calibrated = false; c->r_calibrated = false;      // the loop below overwrites the delayed-scaling tables
  // key: everything baked into the captured nodes (and what a kept calibration belongs to)
  std::string key;
  bool steady = false;
  {
    auto put = [&](const void* q, size_t b) { key.append(static_cast<const char*>(q), b); };
    put(&p->B, 4); put(&p->n_rp, 4); put(&p->n_steps, 4); put(&p->ddim, 4); put(&p->w0, 8); put(&p->w1, 8);
    put(p->t, 4 * p->n_steps); put(p->sqrt_recip, 4 * p->n_steps); put(p->sqrt_recipm1, 4 * p->n_steps);
    if (p->ddim) { put(p->sqrt_a_t, 4 * p->n_steps); put(p->sqrt_1m_a_t, 4 * p->n_steps); put(p->sqrt_a_prev, 4 * p->n_steps); put(p->dir_coef, 4 * p->n_steps); }
    else { put(p->coef1, 4 * p->n_steps); put(p->coef2, 4 * p->n_steps); put(p->stdv, 4 * p->n_steps); put(p->use_noise, 4 * p->n_steps); }
    if (p->apply_apf) put(p->apply_apf, 4 * p->n_steps);
    if (p->noise_scale) put(p->noise_scale, 4 * p->n_steps);
    put(&p->clip_denoised, 4); put(&p->predict_x0, 4); put(&p->n_hard, 4);
    put(&p->philox_sample0, 8); put(&p->philox_total, 8);
    const int has_apf = p->apf.cloud != nullptr; put(&has_apf, 4);
    put(&p->apf.n_points, 4); put(&p->apf.window, 4); put(&p->apf.threshold, 8); put(&p->apf.strength, 8); put(&p->apf.passes, 4);
    const int ch = chain; put(&ch, 4); put(&c->force_x6, 4); put(&p->noise_mode, 4);
  }
  const bool h3 = c->gemm_mode == 2 && !c->force_x6;
  steady = h3 && c->cal_reuse;
  c->s_calibrated = false; c->s_pending = false;
  if (steady) {
    // what the canonical maxima depend on: batch, row variants and their weights, first timestep (the scene and the launch plan invalidate it
    // where they change) -- nothing of the caller's data
    std::string ck;
    auto putc = [&](const void* q, size_t b) { ck.append(static_cast<const char*>(q), b); };
    putc(&p->B, 4); putc(&p->n_rp, 4); putc(p->t, 4); putc(&p->w0, 8); putc(&p->w1, 8);
    const int sp = c->share_prefix; putc(&sp, 4);
    if (!c->c_cal_valid) {                                  // (scene / plan / mode changed: every saved table is stale; its buffer is reused)
      for (auto& kv : c->c_cal_saved) c->c_cal_free.push_back(kv.second);
      c->c_cal_saved.clear();
    }
    if (!c->c_cal_valid || c->c_cal_key != ck) {
      float* t2 = c->obs + 2 * ramp_ctx::MAX_SITES;
      auto it = c->c_cal_saved.find(ck);
      if (it != c->c_cal_saved.end()) {                      // a job shape seen before: its table back into place
        RAMP_HIP_CHECK(hipMemcpyAsync(t2, it->second, ramp_ctx::MAX_SITES * sizeof(float), hipMemcpyDeviceToDevice, s));
      } else {
        CK(canonical_calibration(c, p, s));
        if (c->c_cal_saved.size() < 16) {
          float* keep = nullptr;
          if (!c->c_cal_free.empty()) { keep = c->c_cal_free.back(); c->c_cal_free.pop_back(); }
          else CK(dev_alloc(c, &keep, ramp_ctx::MAX_SITES));
          RAMP_HIP_CHECK(hipMemcpyAsync(keep, t2, ramp_ctx::MAX_SITES * sizeof(float), hipMemcpyDeviceToDevice, s));
          c->c_cal_saved[ck] = keep;
        }
      }
      c->c_cal_valid = true; c->c_cal_key = ck;
      c->launches = 0;
    }
  }
  const int cal_eval = (h3 && c->rerun) ? c->trip_eval : -1;
  c->last_job_steps = h3 ? p->n_steps : 0;
  if (!p->use_graph) {
    CK(sample_body(c, p, s, chain, steady, cal_eval));
  } else {
    if (key != c->graph_key) {
      for (auto& g : c->graph_exec) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
      if (c->graph_rerun) { (void)hipGraphExecDestroy(c->graph_rerun); c->graph_rerun = nullptr; }
      c->graph_key = key; c->graph_rerun_key.clear();
    }
    const int which = steady ? 1 : 0;
    hipGraphExec_t* slot = &c->graph_exec[which];
    if (cal_eval >= 0) {     // the repeat of a flagged job: a graph of its own (kept: a checkpoint that trips at one evaluation does so every job)
      std::string rk = key; rk.append(reinterpret_cast<const char*>(&cal_eval), 4); rk.append(reinterpret_cast<const char*>(&which), 4);
      if (rk != c->graph_rerun_key && c->graph_rerun) { (void)hipGraphExecDestroy(c->graph_rerun); c->graph_rerun = nullptr; }
      c->graph_rerun_key = rk;
      slot = &c->graph_rerun;
    }
    if (!*slot) {
      hipStream_t cs;
      RAMP_HIP_CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
      RAMP_HIP_CHECK(hipStreamSynchronize(s));
      RAMP_HIP_CHECK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
      int rc = sample_body(c, p, cs, chain, steady, cal_eval);
      hipGraph_t g = nullptr;
      hipError_t e = hipStreamEndCapture(cs, &g);
      if (rc != 0) { if (g) (void)hipGraphDestroy(g); (void)hipStreamDestroy(cs); return rc; }
      if (e != hipSuccess) { (void)hipStreamDestroy(cs); RAMP_HIP_CHECK(e); }
      e = hipGraphInstantiate(slot, g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g); (void)hipStreamDestroy(cs);
      RAMP_HIP_CHECK(e);
    }
    RAMP_HIP_CHECK(hipGraphLaunch(*slot, s));
  }
  if (chain_out) RAMP_HIP_CHECK(hipMemcpyAsync(chain_out, c->s_chain, (size_t)(p->n_steps + 1) * n * 4, hipMemcpyDeviceToDevice, s));
  if (x_out) RAMP_HIP_CHECK(hipMemcpyAsync(x_out, c->s_x, n * 4, hipMemcpyDeviceToDevice, s));
  return 0;
}

// ---- receding-horizon replanning --------------------------------------------------------------------
// One replan of DynamicGaussianDiffusionModel.ddim_p_sample_loop (diffusion_model_dynamic.py:533-612) as ONE captured
// graph: q_sample of the current plan -> n_steps x [score, CFG + x0, (last step: static + pursuer APF), DDIM update,
// pinned waypoints] -> smoothing -> collision mask / costs -> selection.  The pursuer's new position is computed by the
// caller BEFORE the replan: the reference hands the environment x[:, stepp, :2], which is the pinned executed state.
static int replan_body(ramp_ctx* c, const ramp_replan_params* p, hipStream_t s, bool calibrate) {
  const int B = p->B, H = c->cfg.horizon, S = c->cfg.state_dim;
  const size_t HS = (size_t)H * S;
  HardConds hc; hc.idx = c->r_hard_idx; hc.val = c->r_hard_val; hc.n = p->n_hard;
  HardConds none;
  CK(launch_replan_init(c->s_x, c->r_xclean, c->r_noise, p->q_sqrt_a, p->q_sqrt_1m_a, c->r_hist, c->r_state, B, H, S, s));
  const bool h3 = c->gemm_mode == 2 && !c->force_x6;
  if (h3) {
    hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(256), 0, s, reinterpret_cast<unsigned*>(c->range_flag), 1);
    RAMP_HIP_CHECK(hipGetLastError());
  }
  for (int j = 0; j < p->n_steps; ++j) {
    const bool last = p->t[j] == 0;                       // the reference's `i == 0`: smoothing + APF on the final step
    if (last) CK(launch_replan_sm(c->s_x, c->r_state, p->sm_window_last, p->sm_dt, p->sm_max_vel, B, H, S, s));
    if (h3) {
      // delayed-scaling tables: evaluation j writes table j & 1, the last one table 3, which the first evaluation of the
      // NEXT replan reads: every steady-state replan sees the same pointers, so its graph is captured once
      const int t_out = j + 1 == p->n_steps ? 3 : (j & 1);
      const int t_in = j == 0 ? 3 : ((j - 1) & 1);
      c->phase = (calibrate && j == 0) ? 1 : 2;
      c->obs_out = c->obs + t_out * ramp_ctx::MAX_SITES;
      c->obs_in = c->obs + t_in * ramp_ctx::MAX_SITES;
      hipLaunchKernelGGL(zero_words_kernel, dim3(ramp_ctx::MAX_SITES / 256), dim3(256), 0, s,
                         reinterpret_cast<unsigned*>(c->obs_out), ramp_ctx::MAX_SITES);
      RAMP_HIP_CHECK(hipGetLastError());
    }
    const float comb[3] = {(float)(1.0 + p->w), -(float)p->w, 0.f};
    const bool shared = p->n_rp > 1 && c->share_prefix;
    const int rc_score = score_all(c, c->s_x, B, p->n_rp, p->t[j], nullptr, c->s_eps, s, shared ? comb : nullptr);
    c->phase = 0;
    CK(rc_score);
    if (c->gemm_mode == 2 && !c->force_x6) {
      hipLaunchKernelGGL(log_flag_kernel, dim3(1), dim3(64), 0, s, c->range_flag, c->trip_log, j);
      RAMP_HIP_CHECK(hipGetLastError());
    }
    CfgMeanArgs m; m.x = c->s_x; m.eps = c->s_eps; m.B = B; m.HS = (int)HS; m.n_rp = shared ? 1 : p->n_rp;
    m.w0 = (float)p->w; m.w1 = 0.f; m.w0p1 = (float)(1.0 + p->w);
    m.sqrt_recip = p->sqrt_recip[j]; m.sqrt_recipm1 = p->sqrt_recipm1[j]; m.clip = p->clip_denoised; m.predict_x0 = p->predict_x0 != 0;
    m.mean = nullptr; m.x0 = c->s_x0;
    CK(launch_cfg_mean(m, s));
    if (last) {
      CK(launch_replan_near(c->s_x, c->r_state, (float)p->thr_pred, c->r_en, B, H, S, s));
      ApfDynArgs a; a.traj = c->s_x0; a.B = B; a.H = H; a.S = S;
      a.points = c->r_static; a.P = p->n_static; a.window = p->window_static; a.affected = H;
      a.thr_query = p->thr_static; a.thr_force = p->thr_static; a.strength = p->strength_static;
      CK(launch_apf_dynamic(a, s));
      a.points = c->r_dyn; a.P = p->n_dyn; a.window = -1; a.affected = H; a.thr_query = p->thr_pred;
      a.strength = p->strength_pred; a.goal = c->s_x + (size_t)(H - 1) * S; a.enable = c->r_en;
      CK(launch_apf_dynamic(a, s));
      CK(launch_replan_goal(c->s_x0, c->s_x, B, H, S, s));
    }
    CK(launch_ddim_finish(c->s_x, c->s_x0, p->sqrt_a_t[j], p->sqrt_1m_a_t[j], p->sqrt_a_prev[j], p->dir_coef[j], none,
                          c->s_x, nullptr, B, H, S, s));
    CK(launch_replan_pin(c->s_x, hc, c->r_hist, c->r_xclean, c->r_state, B, H, S, s));
  }
  CK(launch_replan_sm(c->s_x, c->r_state, p->sm_window_final, p->sm_dt, p->sm_max_vel, B, H, S, s));
  CK(launch_traj_costs(c->s_x, c->r_cost, B, H, S, p->n_cost + p->n_extra, p->cost_thr, c->r_mask, c->r_plen, c->r_smooth, s));
  CK(launch_replan_select(c->s_x, c->r_mask, c->r_plen, c->r_smooth, p->w_smooth, p->w_len, c->r_best, c->r_result, B, H, S, s));
  return 0;
}

int ramp_replan(ramp_ctx* c, const ramp_replan_params* p, const ramp_replan_state* st, float* best_out, float* batch_out,
                int32_t* mask_out, ramp_replan_result* result_host, void* stream) {
  RAMP_REQUIRE(c && p && st && result_host, "null argument");
  RAMP_REQUIRE(c->finalized && c->cross_bias, "context not ready (weights / scene)");
  const int B = p->B, H = c->cfg.horizon, S = c->cfg.state_dim;
  RAMP_REQUIRE(B > 0 && p->n_rp == 2 && p->n_steps >= 1 && p->n_steps <= 64, "bad replan dims");
  RAMP_REQUIRE(p->t && p->sqrt_recip && p->sqrt_recipm1 && p->sqrt_a_t && p->sqrt_1m_a_t && p->sqrt_a_prev && p->dir_coef, "missing schedule arrays");
  RAMP_REQUIRE(p->n_hard >= 0 && p->n_hard <= 16 && (p->n_hard == 0 || (p->hard_idx_host && p->hard_val)), "bad hard conditions");
  RAMP_REQUIRE(p->static_pts && p->n_static > 0 && p->n_dyn > 0 && p->cost_cloud && p->n_cost > 0 && p->n_extra >= 0, "bad clouds");
  RAMP_REQUIRE(st->noise && st->history && st->n_hist >= 1 && st->n_hist <= H && st->stepp >= 0 && st->stepp < H, "bad replan state");
  RAMP_REQUIRE(st->dyn_pts_host && (st->near == 0 || p->n_extra == 0 || st->extra_pts_host), "missing pursuer points");
  RAMP_REQUIRE(st->x_clean || c->r_best, "no current plan: pass x_clean on the first replan");
  for (int j = 0; j < p->n_hard; ++j) RAMP_REQUIRE(p->hard_idx_host[j] >= 0 && p->hard_idx_host[j] < H, "hard index out of range");
  hipStream_t s = as_stream(stream);
  const size_t HS = (size_t)H * S, n = (size_t)B * HS;
  // ---- fixed buffers the graphs read
  CK(ensure_sampler_buffers(c, B, p->n_rp, 0, false));
  bool moved = false;
  if (!c->r_state) {
    float* q; CK(dev_alloc(c, &q, 16)); c->r_state = reinterpret_cast<ReplanState*>(q);
    CK(dev_alloc(c, &q, 8)); c->r_result = reinterpret_cast<int*>(q);
    CK(dev_alloc(c, &q, 16)); c->r_hard_idx = reinterpret_cast<int*>(q);
    CK(dev_alloc(c, &c->r_hist, HS)); CK(dev_alloc(c, &c->r_xclean, HS)); CK(dev_alloc(c, &c->r_best, HS));
    moved = true;
  }
  if ((size_t)B > c->r_cap_B) {
    float* q;
    CK(dev_alloc(c, &c->r_noise, n)); CK(dev_alloc(c, &c->r_plen, B)); CK(dev_alloc(c, &c->r_smooth, B));
    CK(dev_alloc(c, &q, B)); c->r_mask = reinterpret_cast<int*>(q);
    CK(dev_alloc(c, &q, B)); c->r_en = reinterpret_cast<int*>(q);
    CK(dev_alloc(c, &c->r_hard_val, (size_t)16 * B * S));
    c->r_cap_B = B; moved = true;
  }
  if ((size_t)p->n_static > c->r_cap_static) {
    float* q; CK(dev_alloc(c, &q, (size_t)p->n_static * 4)); c->r_static = reinterpret_cast<double*>(q);
    c->r_cap_static = p->n_static; moved = true;
  }
  if ((size_t)p->n_dyn > c->r_cap_dyn) {
    float* q; CK(dev_alloc(c, &q, (size_t)p->n_dyn * 4)); c->r_dyn = reinterpret_cast<double*>(q);
    c->r_cap_dyn = p->n_dyn; moved = true;
  }
  if ((size_t)(p->n_cost + p->n_extra) > c->r_cap_cost) {
    CK(dev_alloc(c, &c->r_cost, (size_t)(p->n_cost + p->n_extra) * 2)); c->r_cap_cost = p->n_cost + p->n_extra; moved = true;
  }
  if (moved) c->r_key.clear();
  // ---- this replan's inputs
  RAMP_HIP_CHECK(hipMemcpyAsync(c->r_noise, st->noise, n * 4, hipMemcpyDeviceToDevice, s));
  RAMP_HIP_CHECK(hipMemcpyAsync(c->r_hist, st->history, (size_t)st->n_hist * S * 4, hipMemcpyDeviceToDevice, s));
  RAMP_HIP_CHECK(hipMemcpyAsync(c->r_xclean, st->x_clean ? st->x_clean : c->r_best, HS * 4, hipMemcpyDeviceToDevice, s));
  RAMP_HIP_CHECK(hipMemcpyAsync(c->r_static, p->static_pts, (size_t)p->n_static * 16, hipMemcpyDeviceToDevice, s));
  RAMP_HIP_CHECK(hipMemcpyAsync(c->r_cost, p->cost_cloud, (size_t)p->n_cost * 8, hipMemcpyDeviceToDevice, s));
  RAMP_HIP_CHECK(hipMemcpyAsync(c->r_dyn, st->dyn_pts_host, (size_t)p->n_dyn * 16, hipMemcpyHostToDevice, s));
  std::vector<float> far((size_t)p->n_extra * 2, 1.0e9f);     // (alive until the synchronisation below)
  if (p->n_extra) {
    // the pursuer's sphere points join the cost cloud only when it is near the evader; otherwise far-away fillers
    RAMP_HIP_CHECK(hipMemcpyAsync(c->r_cost + (size_t)p->n_cost * 2, st->near ? st->extra_pts_host : far.data(),
                                  (size_t)p->n_extra * 8, hipMemcpyHostToDevice, s));
  }
  if (p->n_hard) {
    RAMP_HIP_CHECK(hipMemcpyAsync(c->r_hard_idx, p->hard_idx_host, p->n_hard * 4, hipMemcpyHostToDevice, s));
    RAMP_HIP_CHECK(hipMemcpyAsync(c->r_hard_val, p->hard_val, (size_t)p->n_hard * B * S * 4, hipMemcpyDeviceToDevice, s));
  }
  ReplanState hs{}; hs.n_hist = st->n_hist; hs.stepp = st->stepp; hs.pursuer[0] = st->pursuer[0]; hs.pursuer[1] = st->pursuer[1];
  RAMP_HIP_CHECK(hipMemcpyAsync(c->r_state, &hs, sizeof(hs), hipMemcpyHostToDevice, s));
  c->launches = 0;
  c->score_calibrated = false; c->s_calibrated = false; c->s_pending = false;      // (the replans carry their maxima in table 3: the sampling jobs' canonical table 2 survives)
  const bool h3 = c->gemm_mode == 2 && !c->force_x6;
  auto run = [&](bool calibrate) -> int {
    if (!p->use_graph) return replan_body(c, p, s, calibrate);
    std::string key;
    auto put = [&](const void* q, size_t b) { key.append(static_cast<const char*>(q), b); };
    put(&p->B, 4); put(&p->n_steps, 4); put(&p->w, 8); put(p->t, 4 * p->n_steps); put(p->sqrt_recip, 4 * p->n_steps);
    put(p->sqrt_recipm1, 4 * p->n_steps); put(p->sqrt_a_t, 4 * p->n_steps); put(p->sqrt_1m_a_t, 4 * p->n_steps);
    put(p->sqrt_a_prev, 4 * p->n_steps); put(p->dir_coef, 4 * p->n_steps); put(&p->q_sqrt_a, 4); put(&p->q_sqrt_1m_a, 4);
    put(&p->clip_denoised, 4); put(&p->predict_x0, 4); put(&p->n_hard, 4); put(&p->sm_window_last, 4); put(&p->sm_window_final, 4); put(&p->sm_dt, 4);
    put(&p->sm_max_vel, 4); put(&p->n_static, 4); put(&p->n_dyn, 4); put(&p->thr_static, 8); put(&p->thr_pred, 8);
    put(&p->strength_static, 8); put(&p->strength_pred, 8); put(&p->window_static, 4); put(&p->n_cost, 4); put(&p->n_extra, 4);
    put(&p->cost_thr, 4); put(&p->w_smooth, 4); put(&p->w_len, 4); put(&c->force_x6, 4);
    if (key != c->r_key) {
      for (auto& g : c->r_graph) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
      c->r_key = key;
    }
    const int which = (h3 && !calibrate) ? 1 : 0;
    if (!c->r_graph[which]) {
      hipStream_t cs;
      RAMP_HIP_CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
      RAMP_HIP_CHECK(hipStreamSynchronize(s));
      RAMP_HIP_CHECK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
      int rc = replan_body(c, p, cs, calibrate);
      hipGraph_t g = nullptr;
      hipError_t e = hipStreamEndCapture(cs, &g);
      if (rc != 0) { if (g) (void)hipGraphDestroy(g); (void)hipStreamDestroy(cs); return rc; }
      if (e != hipSuccess) { (void)hipStreamDestroy(cs); RAMP_HIP_CHECK(e); }
      e = hipGraphInstantiate(&c->r_graph[which], g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g); (void)hipStreamDestroy(cs);
      RAMP_HIP_CHECK(e);
    }
    RAMP_HIP_CHECK(hipGraphLaunch(c->r_graph[which], s));
    return 0;
  };
  // a one-step replan has no second table to carry its maxima in (its only evaluation reads and clears table 3): it always
  // calibrates, on the bf16x6 kernels, instead of running fp16x3 from an empty table
  CK(run(h3 && (!c->r_calibrated || p->n_steps == 1)));
  // ---- the one read-back of a replan: {n_free, rank, row} + the range flag
  int back[4] = {0, 0, 0, 0};
  RAMP_HIP_CHECK(hipMemcpyAsync(back, c->r_result, 12, hipMemcpyDeviceToHost, s));
  if (h3) RAMP_HIP_CHECK(hipMemcpyAsync(back + 3, c->range_flag, 4, hipMemcpyDeviceToHost, s));
  RAMP_HIP_CHECK(hipStreamSynchronize(s));
  result_host->fell_back = 0;
  if (h3 && back[3]) {
    // an operand left the range the delayed scaling assumed: repeat THIS replan (same inputs) on the bf16x6 kernels and
    // start the next one with a calibration evaluation
    c->force_x6 = 1;
    int rc = replan_body(c, p, s, false);
    c->force_x6 = 0;
    CK(rc);
    RAMP_HIP_CHECK(hipMemcpyAsync(back, c->r_result, 12, hipMemcpyDeviceToHost, s));
    RAMP_HIP_CHECK(hipStreamSynchronize(s));
    c->r_calibrated = false;
    result_host->fell_back = back[3];
  } else if (h3) {
    c->r_calibrated = true;
  }
  result_host->n_free = back[0]; result_host->best_rank = back[1]; result_host->best_row = back[2];
  if (best_out) RAMP_HIP_CHECK(hipMemcpyAsync(best_out, c->r_best, HS * 4, hipMemcpyDeviceToDevice, s));
  if (batch_out) RAMP_HIP_CHECK(hipMemcpyAsync(batch_out, c->s_x, n * 4, hipMemcpyDeviceToDevice, s));
  if (mask_out) RAMP_HIP_CHECK(hipMemcpyAsync(mask_out, c->r_mask, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
  return 0;
}

/* compute_trajectory_costs + the winner (cost.py:56-88) for a finished batch, e.g. the high-level plan */
int ramp_select_best(const float* traj, int32_t B, int32_t H, int32_t S, const float* cloud, int32_t n_points, float threshold,
                     float w_smooth, float w_len, int32_t* mask, float* path_len, float* smooth, float* best_out,
                     int32_t* result_dev, void* stream) {
  RAMP_REQUIRE(traj && cloud && mask && path_len && smooth && best_out && result_dev, "null argument");
  hipStream_t s = as_stream(stream);
  CK(launch_traj_costs(traj, cloud, B, H, S, n_points, threshold, mask, path_len, smooth, s));
  return launch_replan_select(traj, mask, path_len, smooth, w_smooth, w_len, best_out, result_dev, B, H, S, s);
}

int ramp_select_from_costs(const int32_t* mask, const float* path_len, const float* smooth, int32_t B, float w_smooth, float w_len,
                           int32_t* result_dev, void* stream) {
  RAMP_REQUIRE(mask && path_len && smooth && result_dev && B > 0, "bad arguments");
  return launch_replan_select(nullptr, mask, path_len, smooth, w_smooth, w_len, nullptr, result_dev, B, 1, 1, as_stream(stream));
}

int ramp_replan_costs(ramp_ctx* c, int32_t B, int32_t* mask_out, float* path_len_out, float* smooth_out, void* stream) {
  RAMP_REQUIRE(c && c->r_mask && mask_out && path_len_out && smooth_out && B > 0 && (size_t)B <= c->r_cap_B, "no replan of this size has run on the context");
  hipStream_t s = as_stream(stream);
  RAMP_HIP_CHECK(hipMemcpyAsync(mask_out, c->r_mask, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
  RAMP_HIP_CHECK(hipMemcpyAsync(path_len_out, c->r_plen, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
  RAMP_HIP_CHECK(hipMemcpyAsync(smooth_out, c->r_smooth, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
  return 0;
}

int ramp_debug_read(ramp_ctx* c, const char* kind, const char* module, float* out, int64_t n_floats,
                    int64_t* n_copied, void* stream) {
  RAMP_REQUIRE(c && kind && module && out && n_copied, "null argument");
  auto it = c->dbg.find(std::string(kind) + "/" + module);
  RAMP_REQUIRE(it != c->dbg.end(), std::string("no debug tap '") + kind + "/" + module + "' (debug_taps off or module unknown)");
  const size_t n = std::min<size_t>((size_t)n_floats, it->second.second);
  RAMP_HIP_CHECK(hipMemcpyAsync(out, it->second.first, n * 4, hipMemcpyDeviceToDevice, as_stream(stream)));
  *n_copied = (int64_t)n;
  return 0;
}

int ramp_profile(ramp_ctx* c, int32_t enable) {
  RAMP_REQUIRE(c, "null argument");
  c->prof_on = enable != 0;
  c->prof_used = 0; c->prof_cat.clear(); c->prof_flops.clear(); c->prof_shape.clear();
  return 0;
}
int ramp_profile_read(ramp_ctx* c, double* ms, double* flops, int64_t* count) {
  RAMP_REQUIRE(c && ms && flops && count, "null argument");
  RAMP_HIP_CHECK(hipDeviceSynchronize());
  for (int i = 0; i < CAT_N; ++i) { ms[i] = 0; flops[i] = 0; count[i] = 0; }
  for (size_t i = 0; i < c->prof_cat.size(); ++i) {
    float t = 0.f;
    RAMP_HIP_CHECK(hipEventElapsedTime(&t, c->prof_ev[2 * i], c->prof_ev[2 * i + 1]));
    ms[c->prof_cat[i]] += t; flops[c->prof_cat[i]] += c->prof_flops[i]; count[c->prof_cat[i]]++;
  }
  if (c->prof_dump) {      // per-shape GEMM table on stderr
    std::map<std::array<int, 4>, std::pair<double, int>> agg;
    for (size_t i = 0; i < c->prof_cat.size(); ++i) {
      if (c->prof_cat[i] != CAT_GEMM) continue;
      float t = 0.f; (void)hipEventElapsedTime(&t, c->prof_ev[2 * i], c->prof_ev[2 * i + 1]);
      auto& e = agg[c->prof_shape[i]]; e.first += t; e.second++;
    }
    fprintf(stderr, "[ramp profile] %8s %5s %5s %4s %6s %10s %9s %8s\n", "M", "N", "K", "taps", "calls", "total_ms", "avg_us", "TFLOP/s");
    for (auto& kv : agg) {
      const auto& sh = kv.first; const double fl = 2.0 * sh[0] * sh[1] * sh[2] * sh[3];
      fprintf(stderr, "[ramp profile] %8d %5d %5d %4d %6d %10.2f %9.1f %8.1f\n", sh[0], sh[1], sh[2], sh[3], kv.second.second,
              kv.second.first, kv.second.first * 1e3 / kv.second.second, fl * kv.second.second / (kv.second.first * 1e-3) / 1e12);
    }
  }
  return 0;
}

// per-kernel sums of the profiled launches (the shape codes the launch wrappers record): index 0 ffx forward, 1 ffx backward,
// 2 tkl (LN1 -> QKV, d(o), out-projection), 3 tklb, 4 ato, 5 abl, 6 tkc, 7 tkw (wide k = 5 convolutions, GroupNorm fused),
// 8 every other launch of the GEMM class (tile kernels, exact-fp32 layers, the round-2 fused forward)
int ramp_profile_read_kernels(ramp_ctx* c, int32_t n, double* ms, double* flops, int64_t* count) {
  RAMP_REQUIRE(c && ms && flops && count && n >= 1 && n <= 16, "bad arguments");
  RAMP_HIP_CHECK(hipDeviceSynchronize());
  for (int i = 0; i < n; ++i) { ms[i] = 0; flops[i] = 0; count[i] = 0; }
  for (size_t i = 0; i < c->prof_cat.size(); ++i) {
    if (c->prof_cat[i] != CAT_GEMM) continue;
    const auto& sh = c->prof_shape[i];
    int k = 8;
    if (sh[1] == -2) k = 0; else if (sh[1] == -3) k = 1;
    else if (sh[3] == -10 || sh[3] == -11) k = 2; else if (sh[3] == -12) k = 3; else if (sh[3] == -4) k = 4; else if (sh[3] == -6) k = 5;
    else if (sh[3] == -5) k = 6; else if (sh[3] == -7) k = 7;
    if (k >= n) continue;
    float t = 0.f;
    RAMP_HIP_CHECK(hipEventElapsedTime(&t, c->prof_ev[2 * i], c->prof_ev[2 * i + 1]));
    ms[k] += t; flops[k] += c->prof_flops[i]; count[k]++;
  }
  return 0;
}

int ramp_set_fallback(ramp_ctx* c, int32_t mode) {
  RAMP_REQUIRE(c, "null argument");
  RAMP_REQUIRE(mode >= 0 && mode <= 2, "ramp_set_fallback: mode 0 (none), 1 (every evaluation on the bf16x6 kernels) or 2 (fp16x3, the flagged evaluation calibrating)");
  RAMP_REQUIRE(mode != 2 || c->trip_eval >= 0, "ramp_set_fallback(ctx, 2): no flagged evaluation on record (ramp_range_status of a flagged ramp_sample first)");
  c->force_x6 = mode == 1 ? 1 : 0;
  c->rerun = mode == 2 ? 1 : 0;
  c->score_calibrated = false; c->r_calibrated = false; c->s_calibrated = false; c->s_pending = false;      // (the canonical table 2 survives: nothing writes it in a bf16x6 job)
  return 0;
}
int ramp_score_mode(ramp_ctx* c, int32_t* mode) {
  RAMP_REQUIRE(c && mode, "null argument");
  *mode = c->score_last_mode;
  return 0;
}
int ramp_range_status(ramp_ctx* c, int32_t* flag, void* stream) {
  RAMP_REQUIRE(c && flag, "null argument");
  *flag = 0;
  if (c->range_flag) {
    RAMP_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
    RAMP_HIP_CHECK(hipMemcpy(flag, c->range_flag, sizeof(int), hipMemcpyDeviceToHost));
    if (!c->rerun) { c->trip_eval = -1; c->trip_site = -1; }      // (a repeat keeps its predecessor's record: what it was built from)
    if (*flag && !c->rerun && c->last_job_steps > 0 && c->trip_log) {      // the first evaluation of the last job that left the guard raised
      std::vector<int> log((size_t)c->last_job_steps);
      RAMP_HIP_CHECK(hipMemcpy(log.data(), c->trip_log, log.size() * sizeof(int), hipMemcpyDeviceToHost));
      for (int j = 0; j < c->last_job_steps; ++j) if (log[j]) { c->trip_eval = j; c->trip_site = log[j] - 1; break; }
    }
  }
  return 0;
}
int ramp_range_trip(ramp_ctx* c, int32_t* eval, int32_t* site) {
  RAMP_REQUIRE(c && eval, "null argument");
  *eval = c->trip_eval;
  if (site) *site = c->trip_site;
  return 0;
}
int ramp_set_calibration_reuse(ramp_ctx* c, int32_t on) {
  RAMP_REQUIRE(c, "null argument");
  c->cal_reuse = on ? 1 : 0;
  c->s_calibrated = false; c->s_pending = false; c->c_cal_valid = false;
  return 0;
}
int ramp_workspace_bytes(ramp_ctx* c, int64_t* bytes) {
  RAMP_REQUIRE(c && bytes, "null argument");
  *bytes = (int64_t)c->arena.total;
  return 0;
}
int ramp_launch_count(ramp_ctx* c, int64_t* n) {
  RAMP_REQUIRE(c && n, "null argument");
  *n = c->launches;
  return 0;
}

}  // extern "C"

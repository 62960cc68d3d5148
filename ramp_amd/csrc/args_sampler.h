// Launch arguments of the sampler, replanning, cost and metric kernels (sampler.hip, metrics.hip).
#pragma once
#include "core.h"

namespace ramp {

// ---- sampler (sampler.hip) ------------------------------------------------------------------
struct CfgMeanArgs {
  const float* x = nullptr;     // (B,H,S)
  const float* eps = nullptr;   // (B*n_rp,H,S) interleaved [v0,v1,(v2)] per trajectory
  float* x0 = nullptr; float* mean = nullptr;   // either may be null
  float* ecomb = nullptr;       // optional
  int B = 0, HS = 0, n_rp = 2;
  float w0 = 0, w1 = 0, w0p1 = 1; // n_rp=2: e=w0p1*v0 - w0*v1 (w0p1 = float(1+w)) ; n_rp=3: e=v2+w0*(v0-v2)+w1*(v1-v2)
  float sqrt_recip = 0, sqrt_recipm1 = 0, coef1 = 0, coef2 = 0; int clip = 1;
  int predict_x0 = 0;           // predict_epsilon=False (the reference constructor's default): the combined network output IS x0
};
int launch_cfg_mean(const CfgMeanArgs& a, hipStream_t s);

struct HardConds { const int* idx = nullptr; const float* val = nullptr; int n = 0; };  // val (n,B,S)

// x = mean + (std * z) * noise_scale ; z = 0 when !use_noise (t == 0) ; then hard conditioning
int launch_ddpm_finish(const float* mean, const float* noise, float stdv, float noise_scale, int use_noise,
                       HardConds hc, float* x, float* chain_out, int B, int H, int S, hipStream_t s);
int launch_ddim_finish(const float* x_in, const float* x0, float sqrt_a_t, float sqrt_1m_a_t, float sqrt_a_prev,
                       float dir_coef, HardConds hc, float* x, float* chain_out, int B, int H, int S, hipStream_t s);
int launch_hard_cond(float* x, HardConds hc, int B, int H, int S, hipStream_t s);
// out[0..n) ~ N(0, 1): Philox4x32-10 + Box-Muller, rec = device {seed, offset in groups of four elements} (sampler.hip)
int launch_philox_normal(float* out, long n, const unsigned long long* rec, hipStream_t s);
// the same stream addressed by GLOBAL sample index: out is this shard's (n_blocks, B, HS) noise block of a job whose whole
// noise block is (n_blocks, B_total, HS); local sample b is global sample sample0 + b, i.e. out[(j B + b) HS + e] = element
// (j B_total + sample0 + b) HS + e of the stream (HS % 4 == 0).  B_total == B, sample0 == 0 is launch_philox_normal.
int launch_philox_normal_sharded(float* out, int n_blocks, int B, int HS, long sample0, long B_total, const unsigned long long* rec, hipStream_t s);

struct ApfArgs {
  float* traj = nullptr;        // (B,H,S) modified in place (xy channels only)
  const float* cloud = nullptr; // (P,2)
  const float* window = nullptr;// (2*win+1) Gaussian weights
  int B = 0, H = 0, S = 0, P = 0, win = 0;
  double thr = 0, strength = 0;
};
int launch_apf(const ApfArgs& a, hipStream_t s);
struct ApfDynArgs {
  float* traj = nullptr;          // (B,H,S) in place (xy only)
  const double* points = nullptr; // (P,2) float64
  const float* goal = nullptr;    // (S) goal state for the pursuer pass, or null
  const int* enable = nullptr;    // (B) per-trajectory switch, or null = all
  int B = 0, H = 0, S = 0, P = 0;
  int window = -1;                // >= 0: static pass around the closest waypoint; < 0: waypoints [0, affected)
  int affected = 0;
  double thr_query = 0, thr_force = 0, strength = 0;
};
int launch_apf_dynamic(const ApfDynArgs& a, hipStream_t s);
// receding-horizon replanning (sampler.hip): what changes from replan to replan, resident on the device
struct ReplanState { int n_hist; int stepp; int pad0; int pad1; float pursuer[2]; float pad2[2]; };
int launch_replan_init(float* x, const float* x_clean, const float* noise, float sa, float s1a, const float* hist,
                       const ReplanState* st, int B, int H, int S, hipStream_t s);
int launch_replan_pin(float* x, HardConds hc, const float* hist, const float* x_clean, const ReplanState* st, int B, int H,
                      int S, hipStream_t s);
int launch_replan_sm(float* x, const ReplanState* st, int window, float dt, float max_vel, int B, int H, int S, hipStream_t s);
int launch_replan_near(const float* x, const ReplanState* st, float thr, int* en, int B, int H, int S, hipStream_t s);
int launch_replan_goal(float* x0, const float* x, int B, int H, int S, hipStream_t s);
int launch_replan_select(const float* traj, const int* mask, const float* plen, const float* smooth, float w_s, float w_l,
                         float* best, int* result, int B, int H, int S, hipStream_t s);
// mask[b] = any_{h,p} ||xy - p|| < thr ; plen[b], smooth[b]
int launch_traj_costs(const float* traj, const float* cloud, int B, int H, int S, int P, float thr,
                      int* mask, float* plen, float* smooth, hipStream_t s);
int launch_traj_metrics(const float* traj, int B, int H, int S, const float* centers, const float* sizes, int n_boxes,
                        float* intensity, float* path_len, float* smooth, hipStream_t s);
// scratch: 2 * H * ceil(B / 256) doubles; out: 1 double
int launch_waypoint_variance(const float* traj, int B, int H, int S, double* scratch, double* out, hipStream_t s);
}  // namespace ramp

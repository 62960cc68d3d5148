// Sampler arithmetic around the score network: classifier-free-guidance combine, x0 prediction,
// posterior mean, DDPM / DDIM update, hard conditioning, artificial potential field, costs.
//
//   p_mean_variance (CFG, x0, clamp, posterior)   diffusion_model_static.py:149-186, 188-229; diffusion_model_3d.py:147-182
//   ddpm_sample_fn                                 sample_functions.py:19-48
//   ddim_p_sample                                  diffusion_model_static.py:259-333
//   apply_hard_conditioning                        sample_functions.py:5-10
//   avoidance (static APF)                         APFhelper.py:37-104
//   collision mask / path length / smoothness      cost.py:3-54
//
// These kernels are HBM-bound elementwise work; every product/sum is written with explicit
// round-to-nearest intrinsics in the reference's own evaluation order so that no FMA contraction
// changes a rounding (the x0 clamp makes borderline elements sensitive to single ulps).
#include "args_sampler.h"

#include <algorithm>

namespace ramp {

__device__ __forceinline__ float mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub(float a, float b) { return __fsub_rn(a, b); }

__global__ __launch_bounds__(256) void cfg_mean_kernel(CfgMeanArgs a) {
  const long n = (long)a.B * a.HS;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const long b = idx / a.HS; const int e = (int)(idx - b * a.HS);
    const float* ep = a.eps + (b * a.n_rp) * a.HS + e;
    float ec;
    if (a.n_rp == 2) {
      // e_comb = (1 + w) * cond - w * uncond          (diffusion_model_static.py:164-165)
      ec = sub(mul(a.w0p1, ep[0]), mul(a.w0, ep[a.HS]));
    } else if (a.n_rp == 3) {
      // e_comb = u + w1 (c1 - u) + w2 (c2 - u)         (diffusion_model_static.py:214)
      const float u = ep[2 * a.HS];
      ec = add(add(u, mul(a.w0, sub(ep[0], u))), mul(a.w1, sub(ep[a.HS], u)));
    } else {
      ec = ep[0];
    }
    const float xv = a.x[idx];
    // predict_start_from_noise (diffusion_model_static.py:109-118): predict_epsilon=False returns the network output itself
    float x0 = a.predict_x0 ? ec : sub(mul(a.sqrt_recip, xv), mul(a.sqrt_recipm1, ec));
    if (a.clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
    if (a.ecomb) a.ecomb[idx] = ec;
    if (a.x0) a.x0[idx] = x0;
    if (a.mean) a.mean[idx] = add(mul(a.coef1, x0), mul(a.coef2, xv));
  }
}
int launch_cfg_mean(const CfgMeanArgs& a, hipStream_t s) {
  RAMP_REQUIRE(a.B > 0 && a.HS > 0 && a.n_rp >= 1 && a.n_rp <= 3, "bad cfg_mean dims");
  const long n = (long)a.B * a.HS;
  long g = (n + 255) / 256; if (g > 8192) g = 8192;
  hipLaunchKernelGGL(cfg_mean_kernel, dim3((int)g), dim3(256), 0, s, a);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

// returns true and the conditioned value when waypoint h of sample b is hard-conditioned
__device__ __forceinline__ bool hard_value(const HardConds& hc, int b, int h, int si, int B, int S, float* out) {
  bool hit = false;
  for (int k = 0; k < hc.n; ++k)            // later entries win, like the reference's dict loop
    if (hc.idx[k] == h) { *out = hc.val[((long)k * B + b) * S + si]; hit = true; }
  return hit;
}

__global__ __launch_bounds__(256) void ddpm_finish_kernel(const float* __restrict__ mean, const float* __restrict__ noise,
                                                           float stdv, float noise_scale,
                                                           int use_noise, HardConds hc, float* __restrict__ x,
                                                           float* __restrict__ chain, int B, int H, int S) {
  const long n = (long)B * H * S;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const int si = (int)(idx % S); const long bh = idx / S; const int h = (int)(bh % H); const int b = (int)(bh / H);
    // x = mean + std * noise * noise_std, noise zeroed at t == 0   (sample_functions.py:34-48)
    const float z = use_noise ? noise[idx] : 0.f;
    float v = add(mean[idx], mul(mul(stdv, z), noise_scale));
    float hv;
    if (hard_value(hc, b, h, si, B, S, &hv)) v = hv;
    x[idx] = v;
    if (chain) chain[idx] = v;
  }
}

__global__ __launch_bounds__(256) void ddim_finish_kernel(const float* __restrict__ x_in, const float* __restrict__ x0,
                                                           float sqrt_a_t, float sqrt_1m_a_t, float sqrt_a_prev,
                                                           float dir_coef, HardConds hc, float* __restrict__ x,
                                                           float* __restrict__ chain, int B, int H, int S) {
  const long n = (long)B * H * S;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const int si = (int)(idx % S); const long bh = idx / S; const int h = (int)(bh % H); const int b = (int)(bh / H);
    // model_output = (x - sqrt(a_t) x0) / sqrt(1 - a_t); x = sqrt(a_prev) x0 + sqrt(1 - a_prev) model_output
    // (diffusion_model_static.py:321-333, eta = 0)
    const float p0 = x0[idx];
    const float mo = __fdiv_rn(sub(x_in[idx], mul(sqrt_a_t, p0)), sqrt_1m_a_t);
    float v = add(mul(sqrt_a_prev, p0), mul(dir_coef, mo));
    float hv;
    if (hard_value(hc, b, h, si, B, S, &hv)) v = hv;
    x[idx] = v;
    if (chain) chain[idx] = v;
  }
}

__global__ __launch_bounds__(256) void hard_cond_kernel(float* __restrict__ x, HardConds hc, int B, int H, int S) {
  const long n = (long)hc.n * B * S;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const int si = (int)(idx % S); const long kb = idx / S; const int b = (int)(kb % B); const int k = (int)(kb / B);
    // duplicate indices: the last one wins
    bool later = false;
    for (int k2 = k + 1; k2 < hc.n; ++k2) later |= (hc.idx[k2] == hc.idx[k]);
    if (!later) x[((long)b * H + hc.idx[k]) * S + si] = hc.val[idx];
  }
}

static inline int ew_grid(long n) { long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }

int launch_ddpm_finish(const float* mean, const float* noise, float stdv, float noise_scale, int use_noise,
                        HardConds hc, float* x, float* chain_out, int B, int H, int S, hipStream_t s) {
  RAMP_REQUIRE(B > 0 && H > 0 && S > 0, "bad dims");
  RAMP_REQUIRE(!use_noise || noise != nullptr, "noise required");
  hipLaunchKernelGGL(ddpm_finish_kernel, dim3(ew_grid((long)B * H * S)), dim3(256), 0, s, mean, noise, stdv,
                     noise_scale, use_noise, hc, x, chain_out, B, H, S);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}
int launch_ddim_finish(const float* x_in, const float* x0, float sqrt_a_t, float sqrt_1m_a_t, float sqrt_a_prev,
                       float dir_coef, HardConds hc, float* x, float* chain_out, int B, int H, int S, hipStream_t s) {
  RAMP_REQUIRE(B > 0 && H > 0 && S > 0, "bad dims");
  hipLaunchKernelGGL(ddim_finish_kernel, dim3(ew_grid((long)B * H * S)), dim3(256), 0, s, x_in, x0, sqrt_a_t,
                     sqrt_1m_a_t, sqrt_a_prev, dir_coef, hc, x, chain_out, B, H, S);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}
int launch_hard_cond(float* x, HardConds hc, int B, int H, int S, hipStream_t s) {
  if (hc.n == 0) return 0;
  hipLaunchKernelGGL(hard_cond_kernel, dim3(ew_grid((long)hc.n * B * S)), dim3(256), 0, s, x, hc, B, H, S);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// APF: one 256-thread block per trajectory.  The cloud streams through LDS in tiles of 1024
// points as double2; each wave owns waypoints w, w+4, ...; lanes split the tile's points and the
// (d^2, index) minimum is reduced with wavefront shuffles (lowest index wins ties, like argmin).
// Distances are float64 like the reference's cKDTree query; directions float32; magnitudes
// float64; forces accumulate hit by hit in waypoint order with a float32 rounding after each add,
// exactly as the reference's sequential `force_field[...] +=` does (APFhelper.py:94-101).
// ------------------------------------------------------------------------------------------
constexpr int APF_TILE = 1024;
constexpr int APF_MAXH = 128;

__global__ __launch_bounds__(256) void apf_kernel(ApfArgs a) {
  __shared__ double2 cl[APF_TILE];
  __shared__ double best_d2[APF_MAXH];
  __shared__ int best_i[APF_MAXH];
  __shared__ float fx[APF_MAXH], fy[APF_MAXH];   // per-hit force (mag * dir), zero if no hit
  __shared__ double fmag[APF_MAXH];
  __shared__ float dirx[APF_MAXH], diry[APF_MAXH];
  __shared__ int hitf[APF_MAXH];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* tr = a.traj + (long)b * a.H * a.S;
  for (int h = tid; h < a.H; h += 256) { best_d2[h] = 1.0e300; best_i[h] = -1; }
  for (int p0 = 0; p0 < a.P; p0 += APF_TILE) {
    const int np = min(APF_TILE, a.P - p0);
    __syncthreads();
    for (int e = tid; e < np; e += 256) cl[e] = make_double2((double)a.cloud[(p0 + e) * 2], (double)a.cloud[(p0 + e) * 2 + 1]);
    __syncthreads();
    for (int h = wave; h < a.H; h += 4) {
      const double qx = (double)tr[h * a.S], qy = (double)tr[h * a.S + 1];
      double bd = 1.0e300; int bi = -1;
      for (int e = lane; e < np; e += 64) {
        const double dx = qx - cl[e].x, dy = qy - cl[e].y;
        const double d2 = dx * dx + dy * dy;
        if (d2 < bd) { bd = d2; bi = p0 + e; }
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        const double od = __shfl_xor(bd, m);
        const int oi = __shfl_xor(bi, m);
        if (od < bd || (od == bd && oi >= 0 && (bi < 0 || oi < bi))) { bd = od; bi = oi; }
      }
      if (lane == 0 && bi >= 0 && (bd < best_d2[h] )) { best_d2[h] = bd; best_i[h] = bi; }
    }
  }
  __syncthreads();
  for (int h = tid; h < a.H; h += 256) {
    const double d = sqrt(best_d2[h]);
    const bool hit = best_i[h] >= 0 && d < a.thr;
    hitf[h] = hit;
    if (hit) {
      const float px = a.cloud[best_i[h] * 2], py = a.cloud[best_i[h] * 2 + 1];
      const float ddx = sub(tr[h * a.S], px), ddy = sub(tr[h * a.S + 1], py);
      const float nrm = sqrtf(add(mul(ddx, ddx), mul(ddy, ddy)));
      const float den = add(nrm, 1e-8f);
      dirx[h] = __fdiv_rn(ddx, den);
      diry[h] = __fdiv_rn(ddy, den);
      fmag[h] = a.strength * exp(-d / a.thr);
    }
  }
  __syncthreads();
  for (int sidx = tid; sidx < a.H; sidx += 256) {
    float Fx = 0.f, Fy = 0.f;
    const int lo = max(0, sidx - a.win), hi = min(a.H - 1, sidx + a.win);
    for (int tau = lo; tau <= hi; ++tau) {
      if (!hitf[tau]) continue;
      const double w = (double)a.window[sidx - tau + a.win];
      Fx = (float)((double)Fx + fmag[tau] * (double)dirx[tau] * w);
      Fy = (float)((double)Fy + fmag[tau] * (double)diry[tau] * w);
    }
    fx[sidx] = Fx; fy[sidx] = Fy;
  }
  __syncthreads();
  for (int h = tid; h < a.H; h += 256) {
    tr[h * a.S] = add(tr[h * a.S], fx[h]);
    tr[h * a.S + 1] = add(tr[h * a.S + 1], fy[h]);
  }
}
int launch_apf(const ApfArgs& a, hipStream_t s) {
  RAMP_REQUIRE(a.B > 0 && a.H > 0 && a.H <= APF_MAXH && a.S >= 2 && a.P > 0 && a.win >= 0, "bad APF dims");
  hipLaunchKernelGGL(apf_kernel, dim3(a.B), dim3(256), 0, s, a);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// Receding-horizon replanning (diffusion_model_dynamic.py:495-624): the elementwise pieces between the score
// evaluations of one replan.  What changes from replan to replan (how many waypoints have been executed, the current
// waypoint, the pursuer's position) lives in a small device record, so the captured graph of one replan never changes.
// ------------------------------------------------------------------------------------------
// x = q_sample(repeat(x_clean), t) = sqrt_ac[t] * x_clean + sqrt_1m_ac[t] * noise (diffusion_model_dynamic.py:671-680),
// then x[:, 0, 2:] = 0, the executed history, and the goal waypoint of the clean plan (:540-546)
__global__ __launch_bounds__(256) void replan_init_kernel(float* __restrict__ x, const float* __restrict__ x_clean,
                                                           const float* __restrict__ noise, float sa, float s1a,
                                                           const float* __restrict__ hist, const ReplanState* __restrict__ st,
                                                           int B, int H, int S) {
  const int n_hist = st->n_hist;
  const long n = (long)B * H * S;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const int si = (int)(idx % S); const int h = (int)((idx / S) % H);
    float v = add(mul(sa, x_clean[h * S + si]), mul(s1a, noise[idx]));
    if (h == 0 && si >= 2) v = 0.f;
    if (h < n_hist) v = hist[h * S + si];
    if (h == H - 1) v = x_clean[h * S + si];
    x[idx] = v;
  }
}
// after every DDIM step: apply_hard_conditioning, then the executed history, the goal of the clean plan, x[:, 0, 2:] = 0
// (diffusion_model_dynamic.py:563-568); one thread per (trajectory, state component), the pinned waypoints in order
__global__ __launch_bounds__(256) void replan_pin_kernel(float* __restrict__ x, HardConds hc, const float* __restrict__ hist,
                                                          const float* __restrict__ x_clean, const ReplanState* __restrict__ st,
                                                          int B, int H, int S) {
  const int n_hist = st->n_hist;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * S) return;
  const int b = idx / S, si = idx - b * S;
  float* xb = x + (long)b * H * S;
  for (int k = 0; k < hc.n; ++k) xb[hc.idx[k] * S + si] = hc.val[((long)k * B + b) * S + si];
  for (int h = 0; h < n_hist; ++h) xb[h * S + si] = hist[h * S + si];
  xb[(H - 1) * S + si] = x_clean[(H - 1) * S + si];
  if (si >= 2) xb[si] = 0.f;
}
// sm(): velocity-limited straight-line states between waypoints stepp and stepp + window, written to
// stepp + 1 .. stepp + window (diffusion_model_dynamic.py:192-214), torch's fp32 evaluation order
__global__ __launch_bounds__(256) void replan_sm_kernel(float* __restrict__ x, const ReplanState* __restrict__ st, int window,
                                                         float dt, float max_vel, int B, int H, int S) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const int stepp = st->stepp;
  if (stepp + window >= H) return;                        // the reference would index past the horizon here
  float* xb = x + (long)b * H * S;
  const float* s1 = xb + stepp * S; const float* s2 = xb + (stepp + window) * S;
  const float dx = sub(s2[0], s1[0]), dy = sub(s2[1], s1[1]);
  const float dist = sqrtf(add(mul(dx, dx), mul(dy, dy)));
  const float ux = dist > 1e-6f ? __fdiv_rn(dx, dist) : 0.f, uy = dist > 1e-6f ? __fdiv_rn(dy, dist) : 0.f;
  const float span = (float)((double)window * (double)dt);          // python: num_steps * dt, then one fp32 division
  const float vx0 = __fdiv_rn(dx, span), vy0 = __fdiv_rn(dy, span);
  const bool fast = sqrtf(add(mul(vx0, vx0), mul(vy0, vy0))) > max_vel;
  const float vx = fast ? mul(ux, max_vel) : vx0, vy = fast ? mul(uy, max_vel) : vy0;
  for (int j = 1; j <= window; ++j) {
    const float tt = mul((float)j, dt);
    float* o = xb + (stepp + j) * S;
    o[0] = add(s1[0], mul(tt, vx)); o[1] = add(s1[1], mul(tt, vy));
    o[2] = vx; o[3] = vy;
  }
}
// en[b] = || x[b, stepp, :2] - pursuer || < thr   (diffusion_model_dynamic.py:414-420: which trajectories get the
// pursuer pass); x0[:, -1] = x[:, -1] afterwards is replan_goal_kernel
__global__ __launch_bounds__(256) void replan_near_kernel(const float* __restrict__ x, const ReplanState* __restrict__ st,
                                                           float thr, int* __restrict__ en, int B, int H, int S) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const float* q = x + ((long)b * H + st->stepp) * S;
  const float dx = sub(q[0], st->pursuer[0]), dy = sub(q[1], st->pursuer[1]);
  en[b] = sqrtf(add(mul(dx, dx), mul(dy, dy))) < thr;
}
__global__ __launch_bounds__(256) void replan_goal_kernel(float* __restrict__ x0, const float* __restrict__ x, int B, int H, int S) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * S) return;
  const int b = idx / S, si = idx - b * S;
  x0[((long)b * H + H - 1) * S + si] = x[((long)b * H + H - 1) * S + si];
}
// compute_trajectory_costs (cost.py:56-88) over the B per-trajectory scalars: min-max normalisation over the
// collision-free ones, total = w_s * smooth + w_l * length, first minimum; the winner is gathered with x[0, 2:] = 0
// (diffusion_model_dynamic.py:607).  One block.  result: {n_free, rank of the winner among the free ones, its row, 0}
__global__ __launch_bounds__(1024) void replan_select_kernel(const float* __restrict__ traj, const int* __restrict__ mask,
                                                              const float* __restrict__ plen, const float* __restrict__ smooth,
                                                              float w_s, float w_l, float* __restrict__ best,
                                                              int* __restrict__ result, int B, int H, int S) {
  __shared__ float r_min[2][16], r_max[2][16];
  __shared__ float r_tot[16]; __shared__ int r_idx[16], r_cnt[16];
  __shared__ float s_lo[2], s_hi[2]; __shared__ int s_best, s_free;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float lo0 = INFINITY, hi0 = -INFINITY, lo1 = INFINITY, hi1 = -INFINITY; int cnt = 0;
  for (int b = tid; b < B; b += 1024)
    if (!mask[b]) { lo0 = fminf(lo0, plen[b]); hi0 = fmaxf(hi0, plen[b]); lo1 = fminf(lo1, smooth[b]); hi1 = fmaxf(hi1, smooth[b]); ++cnt; }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    lo0 = fminf(lo0, __shfl_xor(lo0, m)); hi0 = fmaxf(hi0, __shfl_xor(hi0, m));
    lo1 = fminf(lo1, __shfl_xor(lo1, m)); hi1 = fmaxf(hi1, __shfl_xor(hi1, m)); cnt += __shfl_xor(cnt, m);
  }
  if (lane == 0) { r_min[0][wave] = lo0; r_max[0][wave] = hi0; r_min[1][wave] = lo1; r_max[1][wave] = hi1; r_cnt[wave] = cnt; }
  __syncthreads();
  if (tid == 0) {
    float a0 = INFINITY, b0 = -INFINITY, a1 = INFINITY, b1 = -INFINITY; int c = 0;
    for (int w = 0; w < 16; ++w) { a0 = fminf(a0, r_min[0][w]); b0 = fmaxf(b0, r_max[0][w]); a1 = fminf(a1, r_min[1][w]); b1 = fmaxf(b1, r_max[1][w]); c += r_cnt[w]; }
    s_lo[0] = a0; s_hi[0] = b0; s_lo[1] = a1; s_hi[1] = b1; s_free = c;
  }
  __syncthreads();
  const float pl_lo = s_lo[0], pl_rng = sub(s_hi[0], s_lo[0]), sm_lo = s_lo[1], sm_rng = sub(s_hi[1], s_lo[1]);
  float bt = INFINITY; int bi = 0x7fffffff;
  for (int b = tid; b < B; b += 1024) {
    if (mask[b]) continue;
    const float pl = __fdiv_rn(sub(plen[b], pl_lo), pl_rng), sm = __fdiv_rn(sub(smooth[b], sm_lo), sm_rng);
    const float tot = add(mul(w_s, sm), mul(w_l, pl));
    // a NaN total (0 / 0 when every free trajectory ties) ranks first, as in torch.argmin
    const bool better = (tot != tot) ? !(bt != bt) || b < bi : (!(bt != bt) && (tot < bt || (tot == bt && b < bi)));
    if (bi == 0x7fffffff || better) { bt = tot; bi = b; }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const float ot = __shfl_xor(bt, m); const int oi = __shfl_xor(bi, m);
    if (oi == 0x7fffffff) continue;
    const bool better = (ot != ot) ? (!(bt != bt) || oi < bi) : (!(bt != bt) && (ot < bt || (ot == bt && oi < bi)));
    if (bi == 0x7fffffff || better) { bt = ot; bi = oi; }
  }
  if (lane == 0) { r_tot[wave] = bt; r_idx[wave] = bi; }
  __syncthreads();
  if (tid == 0) {
    float t = INFINITY; int i = 0x7fffffff;
    for (int w = 0; w < 16; ++w) {
      const float ot = r_tot[w]; const int oi = r_idx[w];
      if (oi == 0x7fffffff) continue;
      const bool better = (ot != ot) ? (!(t != t) || oi < i) : (!(t != t) && (ot < t || (ot == t && oi < i)));
      if (i == 0x7fffffff || better) { t = ot; i = oi; }
    }
    s_best = i;
  }
  __syncthreads();
  const int bb = s_best;
  if (bb == 0x7fffffff) { if (tid == 0) { result[0] = 0; result[1] = -1; result[2] = -1; } return; }
  int rank = 0;
  for (int b = tid; b < bb; b += 1024) rank += !mask[b];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) rank += __shfl_xor(rank, m);
  if (lane == 0) r_cnt[wave] = rank;
  __syncthreads();
  if (tid == 0) { int r = 0; for (int w = 0; w < 16; ++w) r += r_cnt[w]; result[0] = s_free; result[1] = r; result[2] = bb; }
  if (traj == nullptr) return;                            // selection from gathered costs only (multi-GPU merge): the winner's owner holds the row
  for (int e = tid; e < H * S; e += 1024) {
    float v = traj[(long)bb * H * S + e];
    if (e < S && e >= 2) v = 0.f;
    best[e] = v;
  }
}

int launch_replan_init(float* x, const float* x_clean, const float* noise, float sa, float s1a, const float* hist,
                       const ReplanState* st, int B, int H, int S, hipStream_t s) {
  hipLaunchKernelGGL(replan_init_kernel, dim3(ew_grid((long)B * H * S)), dim3(256), 0, s, x, x_clean, noise, sa, s1a, hist, st, B, H, S);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}
int launch_replan_pin(float* x, HardConds hc, const float* hist, const float* x_clean, const ReplanState* st, int B, int H,
                      int S, hipStream_t s) {
  hipLaunchKernelGGL(replan_pin_kernel, dim3((B * S + 255) / 256), dim3(256), 0, s, x, hc, hist, x_clean, st, B, H, S);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}
int launch_replan_sm(float* x, const ReplanState* st, int window, float dt, float max_vel, int B, int H, int S, hipStream_t s) {
  RAMP_REQUIRE(S >= 4 && window >= 1, "sm needs (x, y, vx, vy) states");
  hipLaunchKernelGGL(replan_sm_kernel, dim3((B + 255) / 256), dim3(256), 0, s, x, st, window, dt, max_vel, B, H, S);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}
int launch_replan_near(const float* x, const ReplanState* st, float thr, int* en, int B, int H, int S, hipStream_t s) {
  hipLaunchKernelGGL(replan_near_kernel, dim3((B + 255) / 256), dim3(256), 0, s, x, st, thr, en, B, H, S);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}
int launch_replan_goal(float* x0, const float* x, int B, int H, int S, hipStream_t s) {
  hipLaunchKernelGGL(replan_goal_kernel, dim3((B * S + 255) / 256), dim3(256), 0, s, x0, x, B, H, S);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}
int launch_replan_select(const float* traj, const int* mask, const float* plen, const float* smooth, float w_s, float w_l,
                         float* best, int* result, int B, int H, int S, hipStream_t s) {
  hipLaunchKernelGGL(replan_select_kernel, dim3(1), dim3(1024), 0, s, traj, mask, plen, smooth, w_s, w_l, best, result, B, H, S);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// Dynamic-planner APF (APFhelper_dynamic.py:107-142), one block per trajectory, points in float64 like the
// reference's numpy clouds.  window >= 0: static pass, pushes only waypoints [ci - w, min(H-1, ci + w)) around the
// waypoint ci nearest to the cloud; window < 0: pursuer pass over waypoints [0, affected) with the 0.9 / 0.1
// avoid / goal blend.  Each waypoint update is independent (no accumulation across waypoints).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void apf_dyn_kernel(ApfDynArgs a) {
  __shared__ double2 cl[APF_TILE];
  __shared__ double best_d2[APF_MAXH];
  __shared__ int best_i[APF_MAXH];
  __shared__ int s_lo, s_hi;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (a.enable && !a.enable[b]) return;
  float* tr = a.traj + (long)b * a.H * a.S;
  const int nq = a.window >= 0 ? a.H : min(a.affected, a.H);
  for (int h = tid; h < a.H; h += 256) { best_d2[h] = 1.0e300; best_i[h] = -1; }
  for (int p0 = 0; p0 < a.P; p0 += APF_TILE) {
    const int np = min(APF_TILE, a.P - p0);
    __syncthreads();
    for (int e = tid; e < np; e += 256) cl[e] = make_double2(a.points[(p0 + e) * 2], a.points[(p0 + e) * 2 + 1]);
    __syncthreads();
    for (int h = wave; h < nq; h += 4) {
      const double qx = (double)tr[h * a.S], qy = (double)tr[h * a.S + 1];
      double bd = 1.0e300; int bi = -1;
      for (int e = lane; e < np; e += 64) {
        const double dx = qx - cl[e].x, dy = qy - cl[e].y;
        const double d2 = dx * dx + dy * dy;
        if (d2 < bd) { bd = d2; bi = p0 + e; }
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        const double od = __shfl_xor(bd, m);
        const int oi = __shfl_xor(bi, m);
        if (od < bd || (od == bd && oi >= 0 && (bi < 0 || oi < bi))) { bd = od; bi = oi; }
      }
      if (lane == 0 && bi >= 0 && bd < best_d2[h]) { best_d2[h] = bd; best_i[h] = bi; }
    }
  }
  __syncthreads();
  if (tid == 0) {
    double bm = 1.0e300; int ci = 0;                     // np.argmin over distances with inf for misses
    for (int h = 0; h < nq; ++h) {
      const double d = sqrt(best_d2[h]);
      if (best_i[h] >= 0 && d < a.thr_query && d < bm) { bm = d; ci = h; }
    }
    if (a.window >= 0) { s_lo = max(0, ci - a.window); s_hi = min(a.H - 1, ci + a.window); }
    else { s_lo = 0; s_hi = nq; }
  }
  __syncthreads();
  for (int h = s_lo + tid; h < s_hi; h += 256) {
    const double d = sqrt(best_d2[h]);
    if (!(best_i[h] >= 0 && d < a.thr_query)) continue;
    const float tx = tr[h * a.S], ty = tr[h * a.S + 1];
    double ax = (double)tx - a.points[best_i[h] * 2], ay = (double)ty - a.points[best_i[h] * 2 + 1];
    const double an = sqrt(ax * ax + ay * ay) + 1e-8;
    ax /= an; ay /= an;
    double cx = ax, cy = ay;
    if (a.goal) {
      float gx = sub(a.goal[0], tx), gy = sub(a.goal[1], ty);
      const float gn = add(sqrtf(add(mul(gx, gx), mul(gy, gy))), 1e-8f);
      gx = __fdiv_rn(gx, gn); gy = __fdiv_rn(gy, gn);
      cx = 0.9 * ax + 0.1 * (double)gx; cy = 0.9 * ay + 0.1 * (double)gy;
      const double cn = sqrt(cx * cx + cy * cy) + 1e-8;
      cx /= cn; cy /= cn;
    }
    const double force = a.strength * exp(-d / a.thr_force);
    tr[h * a.S] = (float)((double)tx + force * cx);
    tr[h * a.S + 1] = (float)((double)ty + force * cy);
  }
}
int launch_apf_dynamic(const ApfDynArgs& a, hipStream_t s) {
  RAMP_REQUIRE(a.B > 0 && a.H > 0 && a.H <= APF_MAXH && a.S >= 2 && a.P > 0 && a.traj && a.points, "bad dynamic-APF dims");
  hipLaunchKernelGGL(apf_dyn_kernel, dim3(a.B), dim3(256), 0, s, a);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// trajectory costs: collision mask against the cloud, path length, smoothness (cost.py:3-54)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void traj_costs_kernel(const float* __restrict__ traj, const float* __restrict__ cloud,
                                                          int H, int S, int P, float thr, int* __restrict__ mask,
                                                          float* __restrict__ plen, float* __restrict__ smooth) {
  __shared__ float2 cl[APF_TILE];
  __shared__ float xy[APF_MAXH * 2];
  __shared__ int any_hit;
  __shared__ float segl[APF_MAXH], segs[APF_MAXH];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* tr = traj + (long)b * H * S;
  if (tid == 0) any_hit = 0;
  for (int h = tid; h < H; h += 256) { xy[2 * h] = tr[h * S]; xy[2 * h + 1] = tr[h * S + 1]; }
  for (int p0 = 0; p0 < P; p0 += APF_TILE) {
    const int np = min(APF_TILE, P - p0);
    __syncthreads();
    for (int e = tid; e < np; e += 256) cl[e] = make_float2(cloud[(p0 + e) * 2], cloud[(p0 + e) * 2 + 1]);
    __syncthreads();
    int hit = 0;
    for (int e = tid; e < np * H; e += 256) {
      const int h = e / np, pi = e - h * np;
      const float dx = sub(xy[2 * h], cl[pi].x), dy = sub(xy[2 * h + 1], cl[pi].y);
      // torch.norm(diff, dim=-1) < thr
      hit |= (sqrtf(add(mul(dx, dx), mul(dy, dy))) < thr);
    }
    if (__any(hit) && (tid & 63) == 0) atomicOr(&any_hit, 1);
  }
  for (int h = tid; h < H - 1; h += 256) {
    const float dx = sub(tr[(h + 1) * S], tr[h * S]), dy = sub(tr[(h + 1) * S + 1], tr[h * S + 1]);
    segl[h] = sqrtf(add(mul(dx, dx), mul(dy, dy)));
    float ss = 0.f;
    for (int c = 2; c < S; ++c) { const float dv = sub(tr[(h + 1) * S + c], tr[h * S + c]); ss = add(ss, mul(dv, dv)); }
    segs[h] = sqrtf(ss);
  }
  __syncthreads();
  if (tid == 0) {
    float pl = 0.f, sm = 0.f;
    for (int h = 0; h < H - 1; ++h) { pl = add(pl, segl[h]); sm = add(sm, segs[h]); }
    mask[b] = any_hit; plen[b] = pl; smooth[b] = sm;
  }
}
int launch_traj_costs(const float* traj, const float* cloud, int B, int H, int S, int P, float thr, int* mask,
                      float* plen, float* smooth, hipStream_t s) {
  RAMP_REQUIRE(B > 0 && H > 1 && H <= APF_MAXH && S >= 2 && P > 0, "bad cost dims");
  hipLaunchKernelGGL(traj_costs_kernel, dim3(B), dim3(256), 0, s, traj, cloud, H, S, P, thr, mask, plen, smooth);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

// ---- Gaussian noise inside the job: counter-based Philox4x32-10 (Salmon et al., SC'11) + Box-Muller --------------------------
// The reference draws torch.randn on the host stream (sample_functions.py:36, diffusion_model_static.py:239); a throughput job
// draws the same N(0, I) inside its captured graph instead.  Element 4 g + j of the stream is output j of
// philox4x32_10(counter = (lo32(g + offset), hi32(g + offset), 0, 0), key = (lo32(seed), hi32(seed))) turned into a normal:
// u = ((r >> 9) + 0.5) * 2^-23 in (0, 1) (exact in fp32); (z0, z1) = sqrt(-2 ln u0) * (cos, sin)(2 pi u1), (z2, z3) likewise from (u2, u3).
// Host-replicable from (seed, offset) alone (tests/util.py restates it in numpy); seed / offset live in a 16-byte device
// record so that a captured graph draws fresh noise on every replay.
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
// grp_row = groups of four per (sample, block) = HS / 4; a shard's local group g = (j B + b) grp_row + q is the stream's group
// (j B_total + sample0 + b) grp_row + q.  grp_row == 0: the flat stream (local group = stream group).
__global__ __launch_bounds__(256) void philox_normal_kernel(float* __restrict__ out, long n, const unsigned long long* __restrict__ rec,
                                                             long grp_row, long B, long sample0, long B_total) {
  const unsigned long long seed = rec[0], offset = rec[1];
  const long n_grp = (n + 3) >> 2;
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < n_grp; g += (long)gridDim.x * 256) {
    long gg = g;
    if (grp_row > 0) {
      const long sb = g / grp_row, q = g - sb * grp_row;      // sb = j B + b
      const long j = sb / B, b = sb - j * B;
      gg = (j * B_total + sample0 + b) * grp_row + q;
    }
    const unsigned long long ctr = (unsigned long long)gg + offset;
    unsigned c[4] = {(unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
    float z[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float u0 = ((float)(c[2 * h] >> 9) + 0.5f) * 1.1920928955078125e-7f;
      const float u1 = ((float)(c[2 * h + 1] >> 9) + 0.5f) * 1.1920928955078125e-7f;
      const float rad = sqrtf(-2.f * logf(u0));
      float sn, cs;
      sincosf(6.283185307179586f * u1, &sn, &cs);
      z[2 * h] = rad * cs; z[2 * h + 1] = rad * sn;
    }
    const long e = g << 2;
    if (e + 3 < n) *reinterpret_cast<float4*>(out + e) = make_float4(z[0], z[1], z[2], z[3]);
    else for (int j = 0; j < 4 && e + j < n; ++j) out[e + j] = z[j];
  }
}
int launch_philox_normal(float* out, long n, const unsigned long long* rec, hipStream_t s) {
  RAMP_REQUIRE(out && rec && n > 0, "philox: null operand");
  RAMP_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "philox: output must be 16-byte aligned");
  const long n_grp = (n + 3) >> 2;
  hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)std::min<long>((n_grp + 255) / 256, 8192)), dim3(256), 0, s, out, n, rec, 0l, 1l, 0l, 1l);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}
int launch_philox_normal_sharded(float* out, int n_blocks, int B, int HS, long sample0, long B_total, const unsigned long long* rec, hipStream_t s) {
  RAMP_REQUIRE(out && rec && n_blocks > 0 && B > 0 && HS > 0 && HS % 4 == 0, "philox: bad shard dims (H S must be a multiple of 4)");
  RAMP_REQUIRE(sample0 >= 0 && B_total >= sample0 + B, "philox: the shard [sample0, sample0 + B) must lie inside the job's B_total samples");
  RAMP_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "philox: output must be 16-byte aligned");
  const long n = (long)n_blocks * B * HS, n_grp = n >> 2;
  hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)std::min<long>((n_grp + 255) / 256, 8192)), dim3(256), 0, s, out, n, rec,
                     (long)(HS / 4), (long)B, sample0, B_total);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace ramp

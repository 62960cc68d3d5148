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
#include "common.h"

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
    float x0 = sub(mul(a.sqrt_recip, xv), mul(a.sqrt_recipm1, ec));
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

}  // namespace ramp

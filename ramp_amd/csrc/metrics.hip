// Solution-quality metrics of a batch of sampled trajectories, on the device
// (scripts/inference/core/metrics.py:8-81 of the reference): AABB collision intensity, path length, velocity
// smoothness per trajectory, and the waypoint variance (variance of ALL B x B entries of the strictly-upper-
// triangular pairwise distance matrix, per waypoint, summed over waypoints).
#include "args_sampler.h"

namespace ramp {

// one wave per trajectory; lane = waypoint (strided when H > 64)
__global__ __launch_bounds__(64)
void traj_metrics_kernel(const float* __restrict__ traj, int B, int H, int S, const float* __restrict__ centers,
                         const float* __restrict__ sizes, int n_boxes, float* __restrict__ intensity,
                         float* __restrict__ path_len, float* __restrict__ smooth) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* t = traj + (long)b * H * S;
  float hits = 0.f, len = 0.f, sm = 0.f;
  for (int h = lane; h < H; h += 64) {
    const float x = t[h * S], y = t[h * S + 1];
    bool in = false;
    for (int k = 0; k < n_boxes; ++k) {
      // metrics.py:68-75: lower = c - s / 2, upper = c + s / 2, inclusive on both sides
      const float cx = centers[2 * k], cy = centers[2 * k + 1], hx = sizes[2 * k] / 2.f, hy = sizes[2 * k + 1] / 2.f;
      in |= (x >= cx - hx) && (x <= cx + hx) && (y >= cy - hy) && (y <= cy + hy);
    }
    hits += in ? 1.f : 0.f;
    if (h + 1 < H) {
      const float dx = t[(h + 1) * S] - x, dy = t[(h + 1) * S + 1] - y;
      len += sqrtf(dx * dx + dy * dy);
      float acc = 0.f;
      for (int d = 2; d < S; ++d) { const float dv = t[(h + 1) * S + d] - t[h * S + d]; acc += dv * dv; }
      sm += sqrtf(acc);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    hits += __shfl_xor(hits, o); len += __shfl_xor(len, o); sm += __shfl_xor(sm, o);
  }
  if (lane == 0) { intensity[b] = hits / (float)H; path_len[b] = len; smooth[b] = sm; }
}

// partial sums over pairs (i in this block's 256 rows, all j > i) of d_ij and d_ij^2 at waypoint blockIdx.y
__global__ __launch_bounds__(256)
void waypoint_pairs_kernel(const float* __restrict__ traj, int B, int H, int S, double* __restrict__ partial) {
  __shared__ float2 tile[256];
  __shared__ double red[2][4];
  const int h = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  float2 pi = {0.f, 0.f};
  if (i < B) pi = {traj[((long)i * H + h) * S], traj[((long)i * H + h) * S + 1]};
  double s1 = 0.0, s2 = 0.0;
  for (int j0 = blockIdx.x * 256; j0 < B; j0 += 256) {     // tiles left of the diagonal hold no pair with j > i
    const int j = j0 + threadIdx.x;
    __syncthreads();
    tile[threadIdx.x] = j < B ? float2{traj[((long)j * H + h) * S], traj[((long)j * H + h) * S + 1]} : float2{0.f, 0.f};
    __syncthreads();
    const int n = min(256, B - j0);
    float a1 = 0.f, a2 = 0.f;                                // a tile's worth in fp32, tiles in fp64
    for (int jj = 0; jj < n; ++jj) {
      const float dx = pi.x - tile[jj].x, dy = pi.y - tile[jj].y;
      const float d2 = dx * dx + dy * dy;
      const bool use = (j0 + jj > i) && (i < B);
      a1 += use ? sqrtf(d2) : 0.f;
      a2 += use ? d2 : 0.f;
    }
    s1 += a1; s2 += a2;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = partial + ((long)h * gridDim.x + blockIdx.x) * 2;
    o[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    o[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}
// var over the N = B*B entries (the B(B+1)/2 zeros of the lower triangle and diagonal included, unbiased), summed over h
__global__ void waypoint_var_kernel(const double* __restrict__ partial, int nblk, int H, int B, double* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double N = (double)B * (double)B;
  double total = 0.0;
  for (int h = 0; h < H; ++h) {
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < nblk; ++k) { s1 += partial[((long)h * nblk + k) * 2]; s2 += partial[((long)h * nblk + k) * 2 + 1]; }
    total += (s2 - s1 * s1 / N) / (N - 1.0);
  }
  out[0] = total;
}

int launch_traj_metrics(const float* traj, int B, int H, int S, const float* centers, const float* sizes, int n_boxes,
                        float* intensity, float* path_len, float* smooth, hipStream_t s) {
  RAMP_REQUIRE(B > 0 && H > 1 && S >= 2 && n_boxes >= 0, "bad metric dims");
  hipLaunchKernelGGL(traj_metrics_kernel, dim3(B), dim3(64), 0, s, traj, B, H, S, centers, sizes, n_boxes, intensity, path_len, smooth);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}
int launch_waypoint_variance(const float* traj, int B, int H, int S, double* scratch, double* out, hipStream_t s) {
  RAMP_REQUIRE(B > 1 && H > 0 && S >= 2, "bad metric dims");
  const int nblk = (B + 255) / 256;
  hipLaunchKernelGGL(waypoint_pairs_kernel, dim3(nblk, H), dim3(256), 0, s, traj, B, H, S, scratch);
  hipLaunchKernelGGL(waypoint_var_kernel, dim3(1), dim3(64), 0, s, scratch, nblk, H, B, out);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace ramp

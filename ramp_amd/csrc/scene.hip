// Scene encoders in HIP (run once per scene): the 2-D ObstacleEncoderSet (obstacle_encoder.py:52-152) and the
// 3-D ObstacleEncoder in eval mode (obstacle_encoder3d.py:5-94).  0.7 GFLOP per scene — small generic kernels,
// the dense linears reuse the exact-fp32 MFMA GEMM of gemm.hip.  All reductions are deterministic.
#include "args_scene.h"

namespace ramp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gelu_s(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float selu_s(float x) {
  return 1.0507009873554804934193349852946f * (x > 0.f ? x : 1.6732632423543772848170429916717f * (expf(x) - 1.f));
}
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int b = 32; b >= 1; b >>= 1) v += __shfl_xor(v, b);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int b = 32; b >= 1; b >>= 1) v = fmaxf(v, __shfl_xor(v, b));
  return v;
}

// ---- 2-D: per-obstacle centre and max |relative coordinate| (obstacle_encoder.py:70-82) ----
__global__ __launch_bounds__(64) void enc2d_prep_kernel(const float* __restrict__ cloud, int Np, float* __restrict__ centers,
                                                        float* __restrict__ maxd) {
  const int o = blockIdx.x, lane = threadIdx.x;
  const float* p = cloud + (long)o * Np * 2;
  float sx = 0.f, sy = 0.f;
  for (int i = lane; i < Np; i += 64) { sx += p[2 * i]; sy += p[2 * i + 1]; }
  const float cx = wsum(sx) / (float)Np, cy = wsum(sy) / (float)Np;
  float m = 0.f;
  for (int i = lane; i < Np; i += 64) m = fmaxf(m, fmaxf(fabsf(p[2 * i] - cx), fabsf(p[2 * i + 1] - cy)));
  m = wmax(m);
  if (lane == 0) { centers[2 * o] = cx; centers[2 * o + 1] = cy; maxd[o] = m; }
}

// ---- 2-D: per token [ GELU(LN(Linear(2->64)(xy))) | PE(centre) | PE(normalised relative position) ] ----
__global__ __launch_bounds__(256) void enc2d_feat_kernel(const float* __restrict__ cloud, const float* __restrict__ centers,
                                                         const float* __restrict__ maxd, const float* __restrict__ div,
                                                         const float* __restrict__ w0, const float* __restrict__ b0,
                                                         const float* __restrict__ g0, const float* __restrict__ be0,
                                                         float* __restrict__ feat, int Np, int T) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const int o = t / Np;
  const float x = cloud[2 * t], y = cloud[2 * t + 1];
  const float cx = centers[2 * o], cy = centers[2 * o + 1];
  float e = w0[2 * lane] * x + w0[2 * lane + 1] * y + b0[lane];
  const float mean = wsum(e) * (1.f / 64.f);
  const float d = e - mean;
  const float rstd = 1.f / sqrtf(wsum(d * d) * (1.f / 64.f) + 1e-5f);
  e = gelu_s(d * rstd * g0[lane] + be0[lane]);
  const float dv = div[lane >> 1];
  const float den = maxd[o] + 1e-8f;
  const float rx = (x - cx) / den, ry = (y - cy) / den;
  float po, pr;
  if (lane & 1) { po = cosf(cx * dv) + cosf(cy * dv); pr = cosf(rx * dv) + cosf(ry * dv); }
  else { po = sinf(cx * dv) + sinf(cy * dv); pr = sinf(rx * dv) + sinf(ry * dv); }
  float* f = feat + (long)t * 192;
  f[lane] = e; f[64 + lane] = po; f[128 + lane] = pr;
}

// ---- LayerNorm over C = 64 (wave per token) with optional GELU ----
__global__ __launch_bounds__(256) void ln64_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                   const float* __restrict__ b, float* __restrict__ y, int T, int act) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const float v = x[(long)t * 64 + lane];
  const float mean = wsum(v) * (1.f / 64.f);
  const float d = v - mean;
  const float rstd = 1.f / sqrtf(wsum(d * d) * (1.f / 64.f) + 1e-5f);
  float o = d * rstd * g[lane] + b[lane];
  if (act == 1) o = gelu_s(o);
  y[(long)t * 64 + lane] = o;
}

// ---- elementwise: y = act(x * scale[c] + shift[c]) (+ add); act 0 none, 1 GELU, 2 SELU ----
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const float* __restrict__ add,
                                                         float* __restrict__ y, long n, int C, int act) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    float v = x[i];
    if (scale) v = v * scale[c] + shift[c];
    if (act == 1) v = gelu_s(v); else if (act == 2) v = selu_s(v);
    if (add) v += add[i];
    y[i] = v;
  }
}

// BatchNorm1d (eval) folded to scale / shift: scale = w / sqrt(var + eps), shift = b - mean * scale
__global__ void bn_fold_kernel(const float* w, const float* b, const float* rm, const float* rv, float* scale, float* shift, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) { const float s = w[c] / sqrtf(rv[c] + 1e-5f); scale[c] = s; shift[c] = b[c] - rm[c] * s; }
}

// ---- tiny-K linear: y[t][n] = b[n] + sum_k x[t][k] W[n][k] ----
__global__ __launch_bounds__(256) void linear_small_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                           const float* __restrict__ b, float* __restrict__ y, long T, int N, int K) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < T * N; i += (long)gridDim.x * 256) {
    const long t = i / N; const int n = (int)(i - t * N);
    float acc = b ? b[n] : 0.f;
    for (int k = 0; k < K; ++k) acc += x[t * K + k] * W[(long)n * K + k];
    y[i] = acc;
  }
}

// ---- segmented column reduce: out[s][c] = mean / max over t < seglen of x[(s*seglen + t)][c] ----
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ x, float* __restrict__ out, int seglen, int C, int mode) {
  const int s = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float* p = x + (long)s * seglen * C + c;
    float acc = mode ? -3.0e38f : 0.f;
    for (int t = 0; t < seglen; ++t) { const float v = p[(long)t * C]; acc = mode ? fmaxf(acc, v) : acc + v; }
    out[(long)s * C + c] = mode ? acc : acc / (float)seglen;
  }
}

// ---- generic softmax attention over T tokens: qkv (T, 3E) [q | k | v], head h at column h*DH; o (T, E) ----
template <int DH>
__global__ __launch_bounds__(64) void attn_generic_kernel(const float* __restrict__ qkv, float* __restrict__ o, int T, int E, float scale) {
  __shared__ float ks[64][DH + 1], vs[64][DH + 1];
  const int h = blockIdx.y, lane = threadIdx.x;
  const int i = blockIdx.x * 64 + lane;
  const bool live = i < T;
  float q[DH], acc[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) { q[d] = live ? qkv[(long)i * 3 * E + h * DH + d] : 0.f; acc[d] = 0.f; }
  float m = -3.0e38f, l = 0.f;
  for (int j0 = 0; j0 < T; j0 += 64) {
    const int nj = min(64, T - j0);
    __syncthreads();
    for (int e = lane; e < nj * DH; e += 64) {
      const int j = e / DH, d = e - j * DH;
      ks[j][d] = qkv[(long)(j0 + j) * 3 * E + E + h * DH + d];
      vs[j][d] = qkv[(long)(j0 + j) * 3 * E + 2 * E + h * DH + d];
    }
    __syncthreads();
    for (int j = 0; j < nj; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) s += q[d] * ks[j][d];
      s *= scale;
      const float mn = fmaxf(m, s);
      const float corr = expf(m - mn), p = expf(s - mn);
      l = l * corr + p;
#pragma unroll
      for (int d = 0; d < DH; ++d) acc[d] = acc[d] * corr + p * vs[j][d];
      m = mn;
    }
  }
  if (live) {
    const float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < DH; ++d) o[(long)i * E + h * DH + d] = acc[d] * inv;
  }
}

static inline int g256(long n) { long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

int scene_enc2d_prep(const float* cloud, int No, int Np, float* centers, float* maxd, hipStream_t s) {
  hipLaunchKernelGGL(enc2d_prep_kernel, dim3(No), dim3(64), 0, s, cloud, Np, centers, maxd);
  RAMP_HIP_CHECK(hipGetLastError()); return 0;
}
int scene_enc2d_feat(const float* cloud, const float* centers, const float* maxd, const float* div, const float* w0,
                     const float* b0, const float* g0, const float* be0, float* feat, int Np, int T, hipStream_t s) {
  hipLaunchKernelGGL(enc2d_feat_kernel, dim3((T + 3) / 4), dim3(256), 0, s, cloud, centers, maxd, div, w0, b0, g0, be0, feat, Np, T);
  RAMP_HIP_CHECK(hipGetLastError()); return 0;
}
int scene_ln64(const float* x, const float* g, const float* b, float* y, int T, int act, hipStream_t s) {
  hipLaunchKernelGGL(ln64_kernel, dim3((T + 3) / 4), dim3(256), 0, s, x, g, b, y, T, act);
  RAMP_HIP_CHECK(hipGetLastError()); return 0;
}
int scene_affine_act(const float* x, const float* scale, const float* shift, const float* add, float* y, long n, int C,
                     int act, hipStream_t s) {
  hipLaunchKernelGGL(affine_act_kernel, dim3(g256(n)), dim3(256), 0, s, x, scale, shift, add, y, n, C, act);
  RAMP_HIP_CHECK(hipGetLastError()); return 0;
}
int scene_bn_fold(const float* w, const float* b, const float* rm, const float* rv, float* scale, float* shift, int C, hipStream_t s) {
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, s, w, b, rm, rv, scale, shift, C);
  RAMP_HIP_CHECK(hipGetLastError()); return 0;
}
int scene_linear_small(const float* x, const float* W, const float* b, float* y, long T, int N, int K, hipStream_t s) {
  hipLaunchKernelGGL(linear_small_kernel, dim3(g256(T * N)), dim3(256), 0, s, x, W, b, y, T, N, K);
  RAMP_HIP_CHECK(hipGetLastError()); return 0;
}
int scene_colreduce(const float* x, float* out, int n_seg, int seglen, int C, int mode, hipStream_t s) {
  hipLaunchKernelGGL(colreduce_kernel, dim3(n_seg), dim3(256), 0, s, x, out, seglen, C, mode);
  RAMP_HIP_CHECK(hipGetLastError()); return 0;
}
int scene_attention(const float* qkv, float* o, int T, int heads, int dh, float scale, hipStream_t s) {
  RAMP_REQUIRE(dh == 16 || dh == 64, "generic attention is instantiated for head dims 16 and 64");
  const dim3 grid((T + 63) / 64, heads);
  if (dh == 16) hipLaunchKernelGGL(attn_generic_kernel<16>, grid, dim3(64), 0, s, qkv, o, T, heads * dh, scale);
  else hipLaunchKernelGGL(attn_generic_kernel<64>, grid, dim3(64), 0, s, qkv, o, T, heads * dh, scale);
  RAMP_HIP_CHECK(hipGetLastError()); return 0;
}

}  // namespace ramp

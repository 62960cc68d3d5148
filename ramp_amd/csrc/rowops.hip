// Row-wise (per trajectory row / per token) kernels of the score network, channels-last fp32.
//
//   GroupNorm(+Mish,+time bias,+residual) fwd / dX   layers.py:280-297, 327-361; layers_attention_mini.py:78-80
//   LayerNorm fwd / dX                                 layers_attention_mini.py:137-139
//   GEGLU fwd / dX                                     layers_attention_mini.py:38-45
//   (4x64 softmax self-attention lives in attention.hip)
//   Downsample1d / Upsample1d fwd / dX                 layers.py:262-277
//   first conv (S -> 32, k5 + 1x1 residual) fwd / dX   layers.py:337-361 for downs.0.0
//   last conv (32 -> S, 1x1) fwd + energy-gradient seed  UnetInference.py:142-145, 26-27
//
// All reductions are deterministic (fixed shuffle / LDS trees, no float atomics).
#include "args_rows.h"

namespace ramp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// mish(x) = x tanh(softplus(x)) (layers.py: nn.Mish).  With e = exp(x): tanh(ln(1 + e)) = n / (n + 2), n = e (e + 2), so
// one v_exp_f32 and one v_rcp_f32 replace expf + log1pf + tanhf (~65 VALU instructions per element, which made the
// GroupNorm kernels VALU-bound at 1.6-2.2 TB/s); no cancellation anywhere, relative error < 1e-6 over the fp32 range
// (e is evaluated at min(x, 20), beyond which the ratio is 1 to fp32 precision).
__device__ __forceinline__ float mish_f(float x) {
  const float e = __builtin_amdgcn_exp2f(fminf(x, 20.f) * 1.44269504088896340736f);
  const float n = e * (e + 2.f);
  return x * (n * __builtin_amdgcn_rcpf(n + 2.f));
}
// d/dx [x r(x)], r = n / (n + 2): r' = 2 n' / (n + 2)^2, n' = 2 e (e + 1)
__device__ __forceinline__ float mish_grad_f(float x) {
  const float e = __builtin_amdgcn_exp2f(fminf(x, 20.f) * 1.44269504088896340736f);
  const float n = e * (e + 2.f);
  const float w = __builtin_amdgcn_rcpf(n + 2.f);
  return n * w + x * (4.f * e * (e + 1.f) * w * w);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
  const float pdf = expf(-0.5f * x * x) * 0.39894228040143267794f;
  return cdf + x * pdf;
}

// ------------------------------------------------------------------------------------------
// GroupNorm: one 256-thread block per row, float4 per thread-slot.  Because 256 % (C/4) == 0 a
// thread always sees the same 4 channels, i.e. one group, decided by lane bits only; partial
// sums are reduced with xor-shuffles over the non-group lane bits, then over the 4 waves in LDS.
// ------------------------------------------------------------------------------------------
constexpr int GN_MAXV = 4;   // float4 slots per thread: supports L*C <= 4096

__device__ __forceinline__ float group_reduce(float v, int gmask, float* red /*[4][8]*/, int g, int lane, int wave) {
#pragma unroll
  for (int b = 1; b < 64; b <<= 1)
    if (!(b & gmask)) v += __shfl_xor(v, b);
  __syncthreads();                       // protect red[] reuse
  if ((lane & ~gmask) == 0) red[wave * 8 + g] = v;
  __syncthreads();
  return red[g] + red[8 + g] + red[16 + g] + red[24 + g];
}

__global__ __launch_bounds__(256) void gn_fwd_kernel(GnArgs a) {
  __shared__ float red[32];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C4 = a.C >> 2, n4 = a.L * C4, cg4 = C4 >> 3;
  const int c4 = tid % C4, g = c4 / cg4;
  const int gmask = (C4 - 1) & ~(cg4 - 1);
  const f32x4* xr = reinterpret_cast<const f32x4*>(a.x + (long)row * a.L * a.C);
  f32x4 v[GN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < GN_MAXV; ++k) {
    const int e = tid + 256 * k;
    if (e < n4) { v[k] = xr[e]; s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]); }
  }
  const float inv_cnt = 1.f / (float)(a.L * (a.C >> 3));
  const float mean = group_reduce(s, gmask, red, g, lane, wave) * inv_cnt;
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < GN_MAXV; ++k) {
    const int e = tid + 256 * k;
    if (e < n4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = v[k][j] - mean; ss += d * d; }
    }
  }
  const float var = group_reduce(ss, gmask, red, g, lane, wave) * inv_cnt;
  const float rstd = 1.f / sqrtf(var + a.eps);
  if (a.stats && (tid < C4) && (c4 % cg4 == 0)) {   // one writer per group
    a.stats[((long)row * 8 + g) * 2 + 0] = mean;
    a.stats[((long)row * 8 + g) * 2 + 1] = rstd;
  }
  const f32x4 gam = reinterpret_cast<const f32x4*>(a.gamma)[c4];
  const f32x4 bet = reinterpret_cast<const f32x4*>(a.beta)[c4];
  f32x4 tb = {0, 0, 0, 0};
  if (a.tbias) tb = reinterpret_cast<const f32x4*>(a.tbias)[c4];
  f32x4* yr = reinterpret_cast<f32x4*>(a.y + (long)row * a.L * a.C);
  const f32x4* rr = a.resid ? reinterpret_cast<const f32x4*>(a.resid + (long)row * a.L * a.C) : nullptr;
#pragma unroll
  for (int k = 0; k < GN_MAXV; ++k) {
    const int e = tid + 256 * k;
    if (e < n4) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float n = (v[k][j] - mean) * rstd * gam[j] + bet[j];
        if (a.mish) n = mish_f(n);
        o[j] = n + tb[j];
      }
      if (rr) { const f32x4 q = rr[e]; o += q; }
      yr[e] = o;
    }
  }
}

__global__ __launch_bounds__(256) void gn_bwd_kernel(GnBwdArgs a) {
  __shared__ float red[32];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C4 = a.C >> 2, n4 = a.L * C4, cg4 = C4 >> 3;
  const int c4 = tid % C4, g = c4 / cg4;
  const int gmask = (C4 - 1) & ~(cg4 - 1);
  const long base = (long)row * a.L * a.C;
  const f32x4* xr = reinterpret_cast<const f32x4*>(a.x + base);
  const f32x4* dyr = reinterpret_cast<const f32x4*>(a.dy + base);
  const float mean = a.stats[((long)row * 8 + g) * 2 + 0];
  const float rstd = a.stats[((long)row * 8 + g) * 2 + 1];
  const f32x4 gam = reinterpret_cast<const f32x4*>(a.gamma)[c4];
  const f32x4 bet = reinterpret_cast<const f32x4*>(a.beta)[c4];
  f32x4 xh[GN_MAXV], gq[GN_MAXV];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < GN_MAXV; ++k) {
    const int e = tid + 256 * k;
    if (e < n4) {
      const f32x4 xv = xr[e], dv = dyr[e];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float h = (xv[j] - mean) * rstd;
        float d = dv[j];
        if (a.mish) d *= mish_grad_f(h * gam[j] + bet[j]);
        d *= gam[j];
        xh[k][j] = h; gq[k][j] = d;
        s1 += d; s2 += d * h;
      }
    }
  }
  const float inv_cnt = 1.f / (float)(a.L * (a.C >> 3));
  const float m1 = group_reduce(s1, gmask, red, g, lane, wave) * inv_cnt;
  const float m2 = group_reduce(s2, gmask, red, g, lane, wave) * inv_cnt;
  f32x4* dxr = reinterpret_cast<f32x4*>(a.dx + base);
  const f32x4* ar = a.add ? reinterpret_cast<const f32x4*>(a.add + base) : nullptr;
#pragma unroll
  for (int k = 0; k < GN_MAXV; ++k) {
    const int e = tid + 256 * k;
    if (e < n4) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (gq[k][j] - m1 - xh[k][j] * m2) * rstd;
      if (ar) { const f32x4 q = ar[e]; o += q; }
      dxr[e] = o;
    }
  }
}

static int gn_check(int R, int L, int C) {
  RAMP_REQUIRE(R > 0 && L > 0, "empty GroupNorm");
  RAMP_REQUIRE(C == 32 || C == 64 || C == 128 || C == 256, "GroupNorm kernel supports C in {32,64,128,256} (8 groups)");
  RAMP_REQUIRE(L * C <= GN_MAXV * 1024, "row too long for the GroupNorm kernel (L*C <= 4096)");
  return 0;
}
int launch_gn_fwd(const GnArgs& a, hipStream_t s) {
  if (int e = gn_check(a.R, a.L, a.C)) return e;
  hipLaunchKernelGGL(gn_fwd_kernel, dim3(a.R), dim3(256), 0, s, a);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}
int launch_gn_bwd(const GnBwdArgs& a, hipStream_t s) {
  if (int e = gn_check(a.R, a.L, a.C)) return e;
  hipLaunchKernelGGL(gn_bwd_kernel, dim3(a.R), dim3(256), 0, s, a);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// LayerNorm over 256 channels: one wave per token, float4 per lane.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int b = 32; b >= 1; b >>= 1) v += __shfl_xor(v, b);
  return v;
}

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ y, int n_tok) {
  const int lane = threadIdx.x & 63;
  const f32x4 gam = reinterpret_cast<const f32x4*>(gamma)[lane];
  const f32x4 bet = reinterpret_cast<const f32x4*>(beta)[lane];
  for (long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6); t < n_tok; t += (long)gridDim.x * 4) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x + t * 256)[lane];
    const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.f / 256.f);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float d = v[j] - mean; ss += d * d; }
    const float rstd = 1.f / sqrtf(wave_sum(ss) * (1.f / 256.f) + 1e-5f);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (v[j] - mean) * rstd * gam[j] + bet[j];
    reinterpret_cast<f32x4*>(y + t * 256)[lane] = o;
  }
}

__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ gamma, const float* __restrict__ add,
                                                      float* __restrict__ dx, int n_tok) {
  const int lane = threadIdx.x & 63;
  const f32x4 gam = reinterpret_cast<const f32x4*>(gamma)[lane];
  for (long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6); t < n_tok; t += (long)gridDim.x * 4) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x + t * 256)[lane];
    const f32x4 d = reinterpret_cast<const f32x4*>(dy + t * 256)[lane];
    const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.f / 256.f);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float q = v[j] - mean; ss += q * q; }
    const float rstd = 1.f / sqrtf(wave_sum(ss) * (1.f / 256.f) + 1e-5f);
    f32x4 xh, gq;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xh[j] = (v[j] - mean) * rstd;
      gq[j] = d[j] * gam[j];
      s1 += gq[j]; s2 += gq[j] * xh[j];
    }
    const float m1 = wave_sum(s1) * (1.f / 256.f), m2 = wave_sum(s2) * (1.f / 256.f);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (gq[j] - m1 - xh[j] * m2) * rstd;
    if (add) { const f32x4 q = reinterpret_cast<const f32x4*>(add + t * 256)[lane]; o += q; }
    reinterpret_cast<f32x4*>(dx + t * 256)[lane] = o;
  }
}

static inline int tok_grid(int n_tok) { int g = (n_tok + 3) / 4; return g < 1 ? 1 : (g > 8192 ? 8192 : g); }

int launch_ln_fwd(const float* x, const float* gamma, const float* beta, float* y, int n_tok, hipStream_t s) {
  RAMP_REQUIRE(n_tok > 0, "empty LayerNorm");
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(tok_grid(n_tok)), dim3(256), 0, s, x, gamma, beta, y, n_tok);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}
int launch_ln_bwd(const float* dy, const float* x, const float* gamma, const float* add, float* dx, int n_tok,
                  hipStream_t s) {
  RAMP_REQUIRE(n_tok > 0, "empty LayerNorm");
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(tok_grid(n_tok)), dim3(256), 0, s, dy, x, gamma, add, dx, n_tok);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// Shared prefix of the rows of one trajectory (engine.hip, net_forward): the n_rp network rows of a trajectory (CFG:
// conditional / unconditional; composition: two scenes + unconditional) get the same x and t and first differ in the
// per-scene constant of the first cross-attention, so everything before it runs once per trajectory.
// expand: out[r] = in[r / n_rp] (+ rowbias[variant(row0 + r)], the cross-attention constant) -- the rows part ways;
// combine: out[b] = sum_j w_j in[b n_rp + j] -- the input gradient is linear in what flows back into the prefix, and
// the sampler only ever uses the weighted sum of the rows' gradients (diffusion_model_static.py:164-165, 214).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void expand_rows_kernel(const float* __restrict__ in, float* __restrict__ out, long n4,
                                                           int row4, int c4, int n_rp, const float* __restrict__ rowbias,
                                                           int rb_stride4, const int* __restrict__ rowvar, int row0) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long r = i / row4; const int w = (int)(i - r * row4);
    f32x4 v = reinterpret_cast<const f32x4*>(in)[(r / n_rp) * row4 + w];
    if (rowbias) v += reinterpret_cast<const f32x4*>(rowbias)[(long)rowvar[row0 + r] * rb_stride4 + (w % c4)];
    reinterpret_cast<f32x4*>(out)[i] = v;
  }
}
int launch_expand_rows(const float* in, float* out, int R, int n_rp, int L, int C, const float* rowbias, int rb_stride,
                       const int* rowvar, int row0, hipStream_t s) {
  RAMP_REQUIRE(R > 0 && n_rp >= 1 && R % n_rp == 0 && C % 4 == 0 && rb_stride % 4 == 0 && (!rowbias || rowvar), "bad expand_rows arguments");
  const long n4 = (long)R * L * C / 4;
  long g = (n4 + 255) / 256; if (g > 16384) g = 16384;
  hipLaunchKernelGGL(expand_rows_kernel, dim3((int)g), dim3(256), 0, s, in, out, n4, L * C / 4, C / 4, n_rp, rowbias,
                     rb_stride / 4, rowvar, row0);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}
__global__ __launch_bounds__(256) void combine_rows_kernel(const float* __restrict__ in, float* __restrict__ out, long n4,
                                                            int row4, int n_rp, float w0, float w1, float w2) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long b = i / row4; const int w = (int)(i - b * row4);
    const f32x4* src = reinterpret_cast<const f32x4*>(in) + (b * n_rp) * row4 + w;
    f32x4 v = src[0] * w0;
    if (n_rp > 1) v += src[row4] * w1;
    if (n_rp > 2) v += src[2 * (long)row4] * w2;
    reinterpret_cast<f32x4*>(out)[i] = v;
  }
}
int launch_combine_rows(const float* in, float* out, int B, int n_rp, int L, int C, const float* w, hipStream_t s) {
  RAMP_REQUIRE(B > 0 && n_rp >= 1 && n_rp <= 3 && C % 4 == 0 && w, "bad combine_rows arguments");
  const long n4 = (long)B * L * C / 4;
  long g = (n4 + 255) / 256; if (g > 16384) g = 16384;
  hipLaunchKernelGGL(combine_rows_kernel, dim3((int)g), dim3(256), 0, s, in, out, n4, L * C / 4, n_rp, w[0],
                     n_rp > 1 ? w[1] : 0.f, n_rp > 2 ? w[2] : 0.f);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// GEGLU
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const float* __restrict__ ag, float* __restrict__ hg,
                                                         long n4, int F4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long t = i / F4; const int j = (int)(i - t * F4);
    const f32x4 av = reinterpret_cast<const f32x4*>(ag)[t * 2 * F4 + j];
    const f32x4 gv = reinterpret_cast<const f32x4*>(ag)[t * 2 * F4 + F4 + j];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = av[e] * gelu_f(gv[e]);
    reinterpret_cast<f32x4*>(hg)[i] = o;
  }
}
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const float* __restrict__ dhg, const float* __restrict__ ag,
                                                         float* __restrict__ dag, long n4, int F4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long t = i / F4; const int j = (int)(i - t * F4);
    const f32x4 av = reinterpret_cast<const f32x4*>(ag)[t * 2 * F4 + j];
    const f32x4 gv = reinterpret_cast<const f32x4*>(ag)[t * 2 * F4 + F4 + j];
    const f32x4 d = reinterpret_cast<const f32x4*>(dhg)[i];
    f32x4 da, dg;
#pragma unroll
    for (int e = 0; e < 4; ++e) { da[e] = d[e] * gelu_f(gv[e]); dg[e] = d[e] * av[e] * gelu_grad_f(gv[e]); }
    reinterpret_cast<f32x4*>(dag)[t * 2 * F4 + j] = da;
    reinterpret_cast<f32x4*>(dag)[t * 2 * F4 + F4 + j] = dg;
  }
}
static inline int ew_grid(long n) { long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g)); }
int launch_geglu_fwd(const float* ag, float* hg, int n_tok, int F, hipStream_t s) {
  RAMP_REQUIRE(n_tok > 0 && F % 4 == 0, "bad GEGLU dims");
  const long n4 = (long)n_tok * (F / 4);
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(ew_grid(n4)), dim3(256), 0, s, ag, hg, n4, F / 4);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}
int launch_geglu_bwd(const float* dhg, const float* ag, float* dag, int n_tok, int F, hipStream_t s) {
  RAMP_REQUIRE(n_tok > 0 && F % 4 == 0, "bad GEGLU dims");
  const long n4 = (long)n_tok * (F / 4);
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(ew_grid(n4)), dim3(256), 0, s, dhg, ag, dag, n4, F / 4);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// stride-2 resampling convolutions (generic gather form, see args_rows.h)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resample_kernel(ResampleArgs a) {
  extern __shared__ float xs[];                         // Lin * Cin
  const int row = blockIdx.x;
  const float* xr = a.x + (long)row * a.Lin * a.Cin;
  for (int e = threadIdx.x; e < a.Lin * a.Cin; e += 256) xs[e] = xr[e];
  __syncthreads();
  const int n_out = a.Lout * a.Cout;
  for (int idx = threadIdx.x; idx < n_out; idx += 256) {
    const int o = idx / a.Cout, co = idx - o * a.Cout;
    float acc = 0.f;
    for (int j = 0; j < a.taps; ++j) {
      int src;
      if (a.mode == 0) src = 2 * o + j - 1;
      else { const int t = o + 1 - j; if (t & 1) continue; src = t >> 1; if (t < 0) continue; }
      if (src < 0 || src >= a.Lin) continue;
      const float* w = a.W + (long)j * a.Cin * a.Cout + co;
      const float* xv = xs + src * a.Cin;
      for (int ci = 0; ci < a.Cin; ++ci) acc += xv[ci] * w[(long)ci * a.Cout];
    }
    if (a.bias) acc += a.bias[co];
    const long oi = (long)row * n_out + idx;
    if (a.add) acc += a.add[oi];
    a.y[oi] = acc;
  }
}
int launch_resample(const ResampleArgs& a, hipStream_t s) {
  RAMP_REQUIRE(a.R > 0 && a.Lin > 0 && a.Lout > 0 && a.Cin > 0 && a.Cout > 0, "bad resample dims");
  const size_t lds = (size_t)a.Lin * a.Cin * sizeof(float);
  RAMP_REQUIRE(lds <= 64 * 1024, "resample row does not fit LDS");
  hipLaunchKernelGGL(resample_kernel, dim3(a.R), dim3(256), lds, s, a);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// first conv block input (Cin = S in {4, 6}: too thin for the MFMA GEMM's float4 staging)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_in_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W5,
                                                           const float* __restrict__ b5, const float* __restrict__ W1,
                                                           const float* __restrict__ b1, float* __restrict__ c1,
                                                           float* __restrict__ res, int n_rp, int H, int S) {
  extern __shared__ float sm[];
  float* xs = sm;                 // H*S
  float* w5 = xs + H * S;         // 5*S*32
  float* w1 = w5 + 5 * S * 32;    // S*32
  const int row = blockIdx.x;
  const float* xr = x + (long)(row / n_rp) * H * S;
  for (int e = threadIdx.x; e < H * S; e += 256) xs[e] = xr[e];
  for (int e = threadIdx.x; e < 5 * S * 32; e += 256) w5[e] = W5[e];
  for (int e = threadIdx.x; e < S * 32; e += 256) w1[e] = W1[e];
  __syncthreads();
  for (int idx = threadIdx.x; idx < H * 32; idx += 256) {
    const int l = idx >> 5, co = idx & 31;
    float acc = 0.f;
    for (int j = 0; j < 5; ++j) {
      const int src = l + j - 2;
      if (src < 0 || src >= H) continue;
      for (int si = 0; si < S; ++si) acc += xs[src * S + si] * w5[(j * S + si) * 32 + co];
    }
    float r = 0.f;
    for (int si = 0; si < S; ++si) r += xs[l * S + si] * w1[si * 32 + co];
    c1[(long)row * H * 32 + idx] = acc + b5[co];
    res[(long)row * H * 32 + idx] = r + b1[co];
  }
}
int launch_conv_in_fwd(const float* x, const float* W5, const float* b5, const float* W1, const float* b1, float* c1,
                       float* res, int R, int n_rp, int H, int S, hipStream_t s) {
  RAMP_REQUIRE(R > 0 && n_rp > 0 && H > 0 && S > 0 && S <= 16, "bad conv_in dims");
  const size_t lds = (size_t)(H * S + 6 * S * 32) * sizeof(float);
  hipLaunchKernelGGL(conv_in_fwd_kernel, dim3(R), dim3(256), lds, s, x, W5, b5, W1, b1, c1, res, n_rp, H, S);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(256) void conv_in_bwd_kernel(const float* __restrict__ dc1, const float* __restrict__ dy,
                                                           const float* __restrict__ W5, const float* __restrict__ W1,
                                                           float* __restrict__ eps, int H, int S) {
  extern __shared__ float sm[];
  float* d1 = sm;                 // H*32
  float* d2 = d1 + H * 32;        // H*32
  float* w5 = d2 + H * 32;        // 5*S*32
  float* w1 = w5 + 5 * S * 32;    // S*32
  const int row = blockIdx.x;
  for (int e = threadIdx.x; e < H * 32; e += 256) {
    d1[e] = dc1[(long)row * H * 32 + e];
    d2[e] = dy[(long)row * H * 32 + e];
  }
  for (int e = threadIdx.x; e < 5 * S * 32; e += 256) w5[e] = W5[e];
  for (int e = threadIdx.x; e < S * 32; e += 256) w1[e] = W1[e];
  __syncthreads();
  for (int idx = threadIdx.x; idx < H * S; idx += 256) {
    const int l = idx / S, si = idx - l * S;
    float acc = 0.f;
    for (int j = 0; j < 5; ++j) {
      const int src = l - j + 2;
      if (src < 0 || src >= H) continue;
      for (int c = 0; c < 32; ++c) acc += d1[src * 32 + c] * w5[(j * S + si) * 32 + c];
    }
    for (int c = 0; c < 32; ++c) acc += d2[l * 32 + c] * w1[si * 32 + c];
    eps[(long)row * H * S + idx] = acc;
  }
}
int launch_conv_in_bwd(const float* dc1, const float* dy, const float* W5, const float* W1, float* eps, int R, int H,
                       int S, hipStream_t s) {
  RAMP_REQUIRE(R > 0 && H > 0 && S > 0 && S <= 16, "bad conv_in dims");
  const size_t lds = (size_t)(2 * H * 32 + 6 * S * 32) * sizeof(float);
  hipLaunchKernelGGL(conv_in_bwd_kernel, dim3(R), dim3(256), lds, s, dc1, dy, W5, W1, eps, H, S);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

// last 1x1 conv + seed of the energy gradient (E = 0.5 sum f^2  =>  dE/df = f)
__global__ __launch_bounds__(256) void conv_out_kernel(const float* __restrict__ a, const float* __restrict__ Wf,
                                                        const float* __restrict__ bf, float* __restrict__ f,
                                                        float* __restrict__ da, int n_tok, int S) {
  __shared__ float w[16 * 32];
  __shared__ float b[16];
  for (int e = threadIdx.x; e < S * 32; e += 256) w[e] = Wf[e];
  if (threadIdx.x < S) b[threadIdx.x] = bf[threadIdx.x];
  __syncthreads();
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_tok) return;
  f32x4 av[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) av[q] = reinterpret_cast<const f32x4*>(a + t * 32)[q];
  float fv[16];
  for (int si = 0; si < S; ++si) {
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc += av[q][e] * w[si * 32 + q * 4 + e];
    fv[si] = acc + b[si];
    if (f) f[t * S + si] = fv[si];
  }
  if (da) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      f32x4 o = {0, 0, 0, 0};
      for (int si = 0; si < S; ++si)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += fv[si] * w[si * 32 + q * 4 + e];
      reinterpret_cast<f32x4*>(da + t * 32)[q] = o;
    }
  }
}
int launch_conv_out(const float* a, const float* Wf, const float* bf, float* f, float* da, int n_tok, int S,
                    hipStream_t s) {
  RAMP_REQUIRE(n_tok > 0 && S > 0 && S <= 16, "bad conv_out dims");
  hipLaunchKernelGGL(conv_out_kernel, dim3((n_tok + 255) / 256), dim3(256), 0, s, a, Wf, bf, f, da, n_tok, S);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// setup kernels
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void time_table_kernel(TimeTableArgs a) {
  __shared__ float e[32], hid[128], st[32];
  const int t = blockIdx.x, tid = threadIdx.x;
  if (tid < 16) {
    // SinusoidalPosEmb(32): freq_j = exp(j * -(ln(1e4)/15)) in fp32, arg = t * freq (layers.py:247-259)
    const float fr = expf((float)tid * (float)(-(9.210340371976184 / 15.0)));
    const float arg = (float)t * fr;
    e[tid] = sinf(arg);
    e[16 + tid] = cosf(arg);
  }
  __syncthreads();
  {
    float acc = 0.f;
    for (int k = 0; k < 32; ++k) acc += e[k] * a.w1[tid * 32 + k];
    hid[tid] = mish_f(acc + a.b1[tid]);
  }
  __syncthreads();
  if (tid < 32) {
    float acc = 0.f;
    for (int k = 0; k < 128; ++k) acc += hid[k] * a.w2[tid * 128 + k];
    const float te = acc + a.b2[tid];
    if (a.temb) a.temb[(long)t * 32 + tid] = te;
    st[tid] = te / (1.f + expf(-te));           // SiLU feeding every cond_mlp (layers.py:340-344)
  }
  __syncthreads();
  for (int i = 0; i < a.n_rtb; ++i) {
    const float* w = a.cond_w[i];
    const float* b = a.cond_b[i];
    for (int c = tid; c < a.couts[i]; c += 128) {
      float acc = 0.f;
      for (int k = 0; k < 32; ++k) acc += st[k] * w[c * 32 + k];
      a.table[(long)t * a.stride + a.offs[i] + c] = acc + b[c];
    }
  }
}
int launch_time_table(const TimeTableArgs& a, hipStream_t s) {
  RAMP_REQUIRE(a.T > 0 && a.n_rtb > 0, "bad time table dims");
  hipLaunchKernelGGL(time_table_kernel, dim3(a.T), dim3(128), 0, s, a);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(256) void cross_bias_kernel(const float* __restrict__ lat, int ctx_dim,
                                                          const float* const* __restrict__ wv,
                                                          const float* const* __restrict__ wo,
                                                          const float* const* __restrict__ bo, int n_blk,
                                                          float* __restrict__ out) {
  __shared__ float ls[512], tmp[256];
  const int v = blockIdx.x / n_blk, blk = blockIdx.x % n_blk, tid = threadIdx.x;
  for (int e = tid; e < ctx_dim; e += 256) ls[e] = lat[(long)v * ctx_dim + e];
  __syncthreads();
  {
    const float* w = wv[blk] + (long)tid * ctx_dim;
    float acc = 0.f;
    for (int k = 0; k < ctx_dim; ++k) acc += ls[k] * w[k];
    tmp[tid] = acc;
  }
  __syncthreads();
  {
    const float* w = wo[blk] + (long)tid * 256;
    float acc = 0.f;
    for (int k = 0; k < 256; ++k) acc += tmp[k] * w[k];
    out[((long)v * n_blk + blk) * 256 + tid] = acc + bo[blk][tid];
  }
}
int launch_cross_bias(const float* lat, int n_var, int ctx_dim, const float* const* wv, const float* const* wo,
                      const float* const* bo, int n_blk, float* out, hipStream_t s) {
  RAMP_REQUIRE(n_var > 0 && n_blk > 0 && ctx_dim > 0 && ctx_dim <= 512, "bad cross-bias dims");
  hipLaunchKernelGGL(cross_bias_kernel, dim3(n_var * n_blk), dim3(256), 0, s, lat, ctx_dim, wv, wo, bo, n_blk, out);
  RAMP_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace ramp

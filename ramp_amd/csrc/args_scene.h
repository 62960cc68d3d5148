// Scene-encoder kernels (scene.hip).
#pragma once
#include "core.h"

namespace ramp {

// ---- scene encoders (scene.hip) ----------------------------------------------------------------
int scene_enc2d_prep(const float* cloud, int No, int Np, float* centers, float* maxd, hipStream_t s);
int scene_enc2d_feat(const float* cloud, const float* centers, const float* maxd, const float* div, const float* w0,
                     const float* b0, const float* g0, const float* be0, float* feat, int Np, int T, hipStream_t s);
int scene_ln64(const float* x, const float* g, const float* b, float* y, int T, int act, hipStream_t s);
int scene_affine_act(const float* x, const float* scale, const float* shift, const float* add, float* y, long n, int C,
                     int act, hipStream_t s);     // act: 0 none, 1 GELU, 2 SELU
int scene_bn_fold(const float* w, const float* b, const float* rm, const float* rv, float* scale, float* shift, int C, hipStream_t s);
int scene_linear_small(const float* x, const float* W, const float* b, float* y, long T, int N, int K, hipStream_t s);
int scene_colreduce(const float* x, float* out, int n_seg, int seglen, int C, int mode /*0 mean, 1 max*/, hipStream_t s);
int scene_attention(const float* qkv, float* o, int T, int heads, int dh, float scale, hipStream_t s);
}  // namespace ramp

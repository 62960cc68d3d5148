/* Diagnostics behind the RAMP sampler's C ABI (include/ramp_hip.h): the per-kernel micro-benchmark and its stress form.  NOT part of the
 * product boundary -- no reference interface corresponds to them (the reference has no micro-benchmarks; its kernels are ATen's).  They are
 * compiled from ramp_amd/csrc/bench.hip into ramp_amd/lib/libramp_hip_tools.so, which also contains every object of libramp_hip.so, and
 * bound by ramp_amd._lib.load_tools() for tests/ (bitwise stress / soak tests of every hand-scheduled kernel) and ramp_amd/tools/. */
#ifndef RAMP_HIP_TOOLS_H
#define RAMP_HIP_TOOLS_H

#include "ramp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* micro-benchmark (profiling tools only): one GEMM shape on the kernel `mode` names (as ramp_op_gemm_mode), operands
 * allocated and filled inside, weights packed once, `warmup` untimed then `iters` timed back-to-back launches on `stream`
 * between two HIP events; *avg_us = microseconds per launch.  flags: 1 bias, 2 residual, 4 GEGLU-forward epilogue
 * (N = 2F), 8 A-multiplier operand (the FF1-dX loader; K = 2 x the operand width), 16 force the 128 x 128 tile,
 * 32 force 3 blocks per CU. */
int ramp_bench_gemm(int32_t M, int32_t N, int32_t K, int32_t taps, int32_t L, int32_t mode, int32_t flags,
                    int32_t warmup, int32_t iters, float* avg_us, void* stream);
/* stress form of the micro-benchmark (tests): `iters` back-to-back launches of the kernel `mode` / `flags` name on the same
 * operands, every launch's output compared bit for bit with the first one's on the device (*mismatching_words: 32-bit words
 * that ever differed; a deterministic kernel gives 0), and -- modes 1..4, rel_err_vs_fp32 non-NULL -- the first output
 * against the exact-fp32 MFMA kernel's on the same operands (max |diff| / max |ref|; -1 where there is no fp32 twin). */
int ramp_stress_gemm(int32_t M, int32_t N, int32_t K, int32_t taps, int32_t L, int32_t mode, int32_t flags, int32_t iters,
                     int64_t* mismatching_words, float* rel_err_vs_fp32, void* stream);

#ifdef __cplusplus
}
#endif

#endif  /* RAMP_HIP_TOOLS_H */

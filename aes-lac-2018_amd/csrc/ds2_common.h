// Shared helpers for the ds2hip kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "ds2_host.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// float4 with only 4-byte alignment: lets hipcc emit one dwordx4 for an unaligned window
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// conv_split.hip (internal): conv2 on the bf16 matrix pipe; 0 = ran, 1 = not selected (the caller runs the direct kernels)
size_t ds2_conv2_split_ws_floats();
size_t ds2_conv2_dgrad_split_ws_floats(int B, int t1);
int ds2_conv2_dgrad_split(const float* d_out, const float* weight, int B, int t1, float* d_in, void* ws, hipStream_t st);
int ds2_conv2_fwd_split(const float* in, const float* weight, const float* bias, int B, int t1, float* out, void* ws,
                        hipStream_t st);

#define DS2_CHECK_LAUNCH()                                                         \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            ds2_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return DS2_ERR_LAUNCH;                                                 \
        }                                                                          \
    } while (0)

#define DS2_HIP(call)                                                              \
    do {                                                                           \
        hipError_t e_ = (call);                                                    \
        if (e_ != hipSuccess) {                                                    \
            ds2_set_error("%s: %s failed: %s", __func__, #call, hipGetErrorString(e_)); \
            return DS2_ERR_LAUNCH;                                                 \
        }                                                                          \
    } while (0)

static inline int ds2_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Environment switches come in two kinds.  SELECTION switches (a kernel family or form that tests and documented A/B runs pick:
// DS2_GRU_FWD / _BWD / _FWD_WIDE / _P2_BF16 / _P2_BF16_BWD, DS2_CONV_SPLIT / _SPLIT_DGRAD / _WGRAD_LDS / _WGRAD_BF16 /
// _DGRAD_ROWS / _DGRAD_MERGE, DS2_GEMM_SPLIT) are read with getenv by every build.  TUNING knobs (tile widths, split-K targets,
// hand-off timing policies, forms no default dispatch reaches: their measured best IS the default, DESIGN.md has the sweeps)
// are read with ds2_tune_env, which is getenv only in the TUNING builds -- `python csrc/build.py --variant tuning`
// (-DDS2_TUNING=1 on every file) and the timing / fault-injection / ablation variants -- and nothing in the release library.
#if defined(DS2_TUNING) || defined(DS2_TIMING) || defined(DS2_FAULT_INJECT) || defined(DS2_ABLATION_BUILD)
#include <stdlib.h>
static inline const char* ds2_tune_env(const char* name) { return getenv(name); }
#else
static inline const char* ds2_tune_env(const char*) { return nullptr; }
#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// Timing-only ablation switches (DS2_GEMM_ABL: gemm.hip, split_bf16.h; DS2_WGRAD_ABL: conv.hip) switch parts of a kernel's
// loop off: the results are WRONG.  They exist for tools/gemm_ablate.py's variant libraries (csrc/build.py:
// build_gemm_variant passes -DDS2_ABLATION_BUILD=1 beside them); a release build that defines one by accident does not compile.
#if ((defined(DS2_GEMM_ABL) && DS2_GEMM_ABL) || (defined(DS2_WGRAD_ABL) && DS2_WGRAD_ABL)) && !defined(DS2_ABLATION_BUILD)
#error "DS2_GEMM_ABL / DS2_WGRAD_ABL produce wrong results: define DS2_ABLATION_BUILD=1 with them (ablation variant libraries only)"
#endif


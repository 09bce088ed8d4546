// Gradient-norm clip + Nesterov-momentum SGD over one flat fp32 buffer (HBM-bound, 16 B/lane).
//
// codes/engine.py:87-90 does clip_grad_norm_(params, max_norm) then optimizer.step(); here the
// clip coefficient is derived on the device from the fp64 sum of squares, so the whole update is
// two launches with no host round trip: traffic = 1 read (norm) + 3 reads + 2 writes (update).
#include "ds2_common.h"

namespace {

constexpr int SS_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, size_t n4, size_t n,
                                                            double* __restrict__ part) {
    float acc = 0.f;
    double dacc = 0.0;
    const size_t stride = (size_t)gridDim.x * 256;
    int cnt = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        if (++cnt == 64) {  // bound the fp32 run length
            dacc += (double)acc;
            acc = 0.f;
            cnt = 0;
        }
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) acc += x[i] * x[i];
    dacc += (double)acc;
    dacc = wave_sum_d(dacc);
    __shared__ double sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = dacc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const double* __restrict__ part, int nparts,
                                                          double* __restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    s = wave_sum_d(s);
    __shared__ double sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ __launch_bounds__(256) void clip_sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ buf, size_t n4, size_t n,
                                                       const double* __restrict__ sumsq, float grad_scale,
                                                       float max_norm, float lr, float momentum, int first_step) {
    float coef = 1.f;
    if (sumsq) {
        const float total = (float)(sqrt(sumsq[0]) * (double)fabsf(grad_scale));
        const float c = max_norm / (total + 1e-6f);
        coef = c < 1.f ? c : 1.f;
    }
    const float gs = grad_scale * coef;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i] * gs;
        const f32x4 bv = first_step ? gv : reinterpret_cast<f32x4*>(buf)[i] * momentum + gv;
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
        pv -= (gv + bv * momentum) * lr;
        reinterpret_cast<f32x4*>(buf)[i] = bv;
        reinterpret_cast<f32x4*>(p)[i] = pv;
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) {
            const float gv = g[i] * gs;
            const float bv = first_step ? gv : buf[i] * momentum + gv;
            p[i] -= lr * (gv + momentum * bv);
            buf[i] = bv;
        }
}

__global__ __launch_bounds__(64) void step_stats_kernel(const float* __restrict__ costs, int B,
                                                        const double* __restrict__ sumsq,
                                                        const uint32_t* const* __restrict__ err_words, int n_err,
                                                        double* __restrict__ out) {
    const int lane = threadIdx.x;
    double s = 0.0, ninf = 0.0;
    for (int i = lane; i < B; i += 64) {
        const float c = costs[i];
        s += (double)c;
        if (isinf(c)) ninf += 1.0;
    }
    s = wave_sum_d(s);
    ninf = wave_sum_d(ninf);
    double e = 0.0;
    for (int i = lane; i < n_err; i += 64)
        if (__hip_atomic_load(err_words[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) e = 1.0;
    e = wave_sum_d(e);
    if (lane == 0) {
        out[0] = s;
        out[1] = sumsq ? sumsq[0] : 0.0;
        out[2] = e > 0.0 ? 1.0 : 0.0;
        out[3] = ninf;
    }
}

}  // namespace

extern "C" int ds2_step_stats(const float* costs, int B, const double* sumsq, const uint32_t* const* err_words,
                              int n_err, double* out, void* stream) {
    DS2_CHECK_ARG(costs && out && B > 0 && n_err >= 0 && (n_err == 0 || err_words));
    hipLaunchKernelGGL(step_stats_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, costs, B, sumsq, err_words, n_err,
                       out);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" size_t ds2_sumsq_ws_bytes(size_t n) {
    (void)n;
    return SS_BLOCKS * sizeof(double);
}

extern "C" int ds2_sumsq(const float* x, size_t n, double* out, void* ws, void* stream) {
    DS2_CHECK_ARG(x && out && ws && n > 0);
    DS2_CHECK_ARG(((uintptr_t)x & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    const size_t n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > SS_BLOCKS) blocks = SS_BLOCKS;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, st, x, n4, n, (double*)ws);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, blocks, out);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" int ds2_clip_sgd_nesterov(float* p, const float* g, float* buf, size_t n, const double* sumsq,
                                     float grad_scale, float max_norm, float lr, float momentum, int first_step,
                                     void* stream) {
    DS2_CHECK_ARG(p && g && buf && n > 0);
    DS2_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)buf) & 15) == 0);
    const size_t n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(clip_sgd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, buf, n4, n, sumsq,
                       grad_scale, max_norm, lr, momentum, first_step);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

// BatchNorm kernels (HBM-bound): conv flavour (B,C,inner) fused with Hardtanh(0,20) and the
// sequence flavour (rows,F) that folds the sum of the two GRU directions into its read.
//
// Statistics: every block accumulates fp32 partial sums over a bounded run (<= a few hundred
// elements per thread), partials are combined in fp64 (NPART per channel) by a finalise kernel
// that also applies the momentum update of the running buffers (unbiased variance), exactly
// as torch.nn.BatchNorm{1,2}d in training mode.
#include "ds2_common.h"

namespace {

constexpr int NPART = 64;

// ----------------------------------------------------------------------------- conv flavour
// grid (NPART, C); x (B,C,inner).  MODE 0: sum x, sum x^2.  MODE 1 (backward): sum g, sum g*xhat
// with g = dy masked by the hardtanh interior 0 < gamma*xhat+beta < 20.
template <int MODE>
__global__ __launch_bounds__(256) void bn2d_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ mean_invstd,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int B, int C, int inner,
                                                          double* __restrict__ part) {
    const int c = blockIdx.y, p = blockIdx.x, tid = threadIdx.x;
    const int chunks = (inner + 1023) / 1024;
    const int items = B * chunks;
    float mu = 0.f, is = 0.f, ga = 0.f, be = 0.f;
    if (MODE == 1) {
        mu = mean_invstd[c];
        is = mean_invstd[C + c];
        ga = gamma[c];
        be = beta[c];
    }
    float s0 = 0.f, s1 = 0.f;
    for (int it = p; it < items; it += NPART) {
        const int b = it / chunks, ch = it % chunks;
        const size_t base = ((size_t)b * C + c) * inner;
        const int i0 = ch * 1024;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 256 + tid;
            if (i < inner) {
                const float v = x[base + i];
                if (MODE == 0) {
                    s0 += v;
                    s1 += v * v;
                } else {
                    const float xh = (v - mu) * is;
                    const float y = ga * xh + be;
                    const float g = (y > 0.f && y < 20.f) ? dy[base + i] : 0.f;
                    s0 += g;
                    s1 += g * xh;
                }
            }
        }
    }
    double d0 = wave_sum_d((double)s0), d1 = wave_sum_d((double)s1);
    __shared__ double sm[4][2];
    if ((tid & 63) == 0) {
        sm[tid >> 6][0] = d0;
        sm[tid >> 6][1] = d1;
    }
    __syncthreads();
    if (tid == 0) {
        part[((size_t)c * NPART + p) * 2 + 0] = sm[0][0] + sm[1][0] + sm[2][0] + sm[3][0];
        part[((size_t)c * NPART + p) * 2 + 1] = sm[0][1] + sm[1][1] + sm[2][1] + sm[3][1];
    }
}

// 16-byte form of the reduction: a (b, c) row of `inner` floats starts at any 4-byte offset, so it is cut into a scalar head
// (up to 3 elements, to the next 16-byte boundary of the tensor), an aligned float4 body and a scalar tail.  Used when the
// tensors themselves are 16-byte aligned (torch allocations are).
template <int MODE>
__global__ __launch_bounds__(256) void bn2d_reduce_vec_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              const float* __restrict__ mean_invstd,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int B, int C, int inner,
                                                              double* __restrict__ part) {
    const int c = blockIdx.y, p = blockIdx.x, tid = threadIdx.x;
    const int chunks = (inner + 1023) / 1024;
    const int items = B * chunks;
    float mu = 0.f, is = 0.f, ga = 0.f, be = 0.f;
    if (MODE == 1) {
        mu = mean_invstd[c];
        is = mean_invstd[C + c];
        ga = gamma[c];
        be = beta[c];
    }
    float s0 = 0.f, s1 = 0.f;
    auto take = [&](float v, float g) {
        if (MODE == 0) {
            s0 += v;
            s1 += v * v;
        } else {
            const float xh = (v - mu) * is;
            const float y = ga * xh + be;
            const float gm = (y > 0.f && y < 20.f) ? g : 0.f;
            s0 += gm;
            s1 += gm * xh;
        }
    };
    for (int it = p; it < items; it += NPART) {
        const int b = it / chunks, ch = it % chunks;
        const size_t base = ((size_t)b * C + c) * inner;
        const int h = min((int)((4 - (base & 3)) & 3), inner);
        const int nq = (inner - h) >> 2;
        const int q = ch * 256 + tid;
        if (q < nq) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + base + h + 4 * (size_t)q);
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
            if (MODE == 1) g = *reinterpret_cast<const f32x4*>(dy + base + h + 4 * (size_t)q);
#pragma unroll
            for (int e = 0; e < 4; ++e) take(v[e], g[e]);
        }
        if (ch == 0) {
            const int tail0 = h + 4 * nq, e = tid < 4 ? tid : tail0 + tid - 4;
            if ((tid < h) || (tid >= 4 && tid < 8 && e < inner)) take(x[base + e], MODE == 1 ? dy[base + e] : 0.f);
        }
    }
    double d0 = wave_sum_d((double)s0), d1 = wave_sum_d((double)s1);
    __shared__ double sm[4][2];
    if ((tid & 63) == 0) {
        sm[tid >> 6][0] = d0;
        sm[tid >> 6][1] = d1;
    }
    __syncthreads();
    if (tid == 0) {
        part[((size_t)c * NPART + p) * 2 + 0] = sm[0][0] + sm[1][0] + sm[2][0] + sm[3][0];
        part[((size_t)c * NPART + p) * 2 + 1] = sm[0][1] + sm[1][1] + sm[2][1] + sm[3][1];
    }
}

// ----------------------------------------------------------------------------- sequence flavour
// grid (NPART, ceil(F/64)); block = 64 columns x 4 row lanes
template <int MODE>
__global__ __launch_bounds__(256) void bn1d_reduce_kernel(const float* __restrict__ xa, const float* __restrict__ xb,
                                                          const float* __restrict__ dy,
                                                          const float* __restrict__ mean_invstd, int rows, int F,
                                                          double* __restrict__ part) {
    const int p = blockIdx.x, tid = threadIdx.x;
    const int col = blockIdx.y * 64 + (tid & 63), ry = tid >> 6;
    float mu = 0.f, is = 0.f;
    if (MODE == 1 && col < F) {
        mu = mean_invstd[col];
        is = mean_invstd[F + col];
    }
    float s0 = 0.f, s1 = 0.f;
    if (col < F) {
        for (int r = p * 4 + ry; r < rows; r += NPART * 4) {
            const size_t o = (size_t)r * F + col;
            float v = xa[o];
            if (xb) v += xb[o];
            if (MODE == 0) {
                s0 += v;
                s1 += v * v;
            } else {
                const float g = dy[o];
                s0 += g;
                s1 += g * (v - mu) * is;
            }
        }
    }
    __shared__ float sm[4][64][2];
    sm[ry][tid & 63][0] = s0;
    sm[ry][tid & 63][1] = s1;
    __syncthreads();
    if (ry == 0 && col < F) {
        const int l = tid & 63;
        const double d0 = (double)sm[0][l][0] + (double)sm[1][l][0] + (double)sm[2][l][0] + (double)sm[3][l][0];
        const double d1 = (double)sm[0][l][1] + (double)sm[1][l][1] + (double)sm[2][l][1] + (double)sm[3][l][1];
        part[((size_t)col * NPART + p) * 2 + 0] = d0;
        part[((size_t)col * NPART + p) * 2 + 1] = d1;
    }
}

// one wave per channel: combine NPART fp64 partials
__global__ __launch_bounds__(64) void bn_finalize_stats_kernel(const double* __restrict__ part, int C, double count,
                                                               float eps, float momentum, int use_running,
                                                               float* __restrict__ running_mean,
                                                               float* __restrict__ running_var,
                                                               float* __restrict__ mean_invstd) {
    const int c = blockIdx.x, lane = threadIdx.x;
    if (use_running) {
        if (lane == 0) {
            mean_invstd[c] = running_mean[c];
            mean_invstd[C + c] = (float)(1.0 / sqrt((double)running_var[c] + (double)eps));
        }
        return;
    }
    double s0 = part[((size_t)c * NPART + lane) * 2 + 0];
    double s1 = part[((size_t)c * NPART + lane) * 2 + 1];
    s0 = wave_sum_d(s0);
    s1 = wave_sum_d(s1);
    if (lane == 0) {
        const double mean = s0 / count;
        double var = s1 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_invstd[c] = (float)mean;
        mean_invstd[C + c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
        }
    }
}

// backward finalise: dgamma = sum g*xhat, dbeta = sum g; coef[c] = mean g, coef[C+c] = mean g*xhat
__global__ __launch_bounds__(64) void bn_finalize_bwd_kernel(const double* __restrict__ part, int C, double count,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             float* __restrict__ coef) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double s0 = part[((size_t)c * NPART + lane) * 2 + 0];
    double s1 = part[((size_t)c * NPART + lane) * 2 + 1];
    s0 = wave_sum_d(s0);
    s1 = wave_sum_d(s1);
    if (lane == 0) {
        dbeta[c] = (float)s0;
        dgamma[c] = (float)s1;
        coef[c] = (float)(s0 / count);
        coef[C + c] = (float)(s1 / count);
    }
}

// y = clamp(gamma*(x-mu)*invstd + beta, 0, 20), same layout
__global__ __launch_bounds__(256) void bn2d_apply_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ mean_invstd,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int C, int inner,
                                                         float* __restrict__ y) {
    const int bc = blockIdx.y, c = bc % C;
    const float sc = gamma[c] * mean_invstd[C + c];
    const float sh = beta[c] - mean_invstd[c] * sc;
    const size_t base = (size_t)bc * inner;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < inner; i += gridDim.x * 256) {
        const float v = x[base + i] * sc + sh;
        y[base + i] = fminf(fmaxf(v, 0.f), 20.f);
    }
}

// x (B,C,D,T) -> y (T,B,C*D) with BN + hardtanh; per b a (CD x T) -> (T x CD) tile transpose
__global__ __launch_bounds__(256) void bn2d_apply_tbf_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ mean_invstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int B, int C, int D,
                                                             int T, float* __restrict__ y) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int CD = C * D;
    const int t0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int f = f0 + ty + i, t = t0 + tx;
        if (f < CD && t < T) {
            const int c = f / D;
            const float sc = gamma[c] * mean_invstd[C + c];
            const float sh = beta[c] - mean_invstd[c] * sc;
            const float v = x[((size_t)b * CD + f) * T + t] * sc + sh;
            tile[ty + i][tx] = fminf(fmaxf(v, 0.f), 20.f);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int t = t0 + ty + i, f = f0 + tx;
        if (f < CD && t < T) y[((size_t)t * B + b) * CD + f] = tile[tx][ty + i];
    }
}

// dx = gamma*invstd*(g - mean_g - xhat*mean_gxhat), g = dy masked by the hardtanh interior
__global__ __launch_bounds__(256) void bn2d_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ mean_invstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ coef, int C, int inner,
                                                             float* __restrict__ dx) {
    const int bc = blockIdx.y, c = bc % C;
    const float mu = mean_invstd[c], is = mean_invstd[C + c], ga = gamma[c], be = beta[c];
    const float mg = coef[c], mgx = coef[C + c];
    const size_t base = (size_t)bc * inner;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < inner; i += gridDim.x * 256) {
        const float xh = (x[base + i] - mu) * is;
        const float yv = ga * xh + be;
        const float g = (yv > 0.f && yv < 20.f) ? dy[base + i] : 0.f;
        dx[base + i] = ga * is * (g - mg - xh * mgx);
    }
}

// 16-byte forms of the two elementwise conv-flavour kernels (row = one (b, c) plane; head / aligned body / tail as above)
template <typename F>
__device__ __forceinline__ void bn2d_row_vec(size_t base, int inner, F&& f4, int bx, int nbx, int tid) {
    const int h = min((int)((4 - (base & 3)) & 3), inner);
    const int nq = (inner - h) >> 2;
    for (int q = bx * 256 + tid; q < nq; q += nbx * 256) f4(base + h + 4 * (size_t)q, 4);
    if (bx == 0) {
        const int tail0 = h + 4 * nq, e = tid < 4 ? tid : tail0 + tid - 4;
        if ((tid < h) || (tid >= 4 && tid < 8 && e < inner)) f4(base + e, 1);
    }
}

__global__ __launch_bounds__(256) void bn2d_bwd_apply_vec_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                 const float* __restrict__ mean_invstd,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta,
                                                                 const float* __restrict__ coef, int C, int inner,
                                                                 float* __restrict__ dx) {
    const int bc = blockIdx.y, c = bc % C;
    const float mu = mean_invstd[c], is = mean_invstd[C + c], ga = gamma[c], be = beta[c];
    const float mg = coef[c], mgx = coef[C + c];
    auto one = [&](float xv, float g) {
        const float xh = (xv - mu) * is;
        const float yv = ga * xh + be;
        const float gm = (yv > 0.f && yv < 20.f) ? g : 0.f;
        return ga * is * (gm - mg - xh * mgx);
    };
    bn2d_row_vec((size_t)bc * inner, inner, [&](size_t o, int n) {
        if (n == 4) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + o), g = *reinterpret_cast<const f32x4*>(dy + o);
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = one(xv[e], g[e]);
            *reinterpret_cast<f32x4*>(dx + o) = r;
        } else {
            dx[o] = one(x[o], dy[o]);
        }
    }, blockIdx.x, gridDim.x, threadIdx.x);
}

__global__ __launch_bounds__(256) void bn2d_apply_vec_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ mean_invstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int C, int inner,
                                                             float* __restrict__ y) {
    const int bc = blockIdx.y, c = bc % C;
    const float sc = gamma[c] * mean_invstd[C + c];
    const float sh = beta[c] - mean_invstd[c] * sc;
    bn2d_row_vec((size_t)bc * inner, inner, [&](size_t o, int n) {
        if (n == 4) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + o);
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = fminf(fmaxf(xv[e] * sc + sh, 0.f), 20.f);
            *reinterpret_cast<f32x4*>(y + o) = r;
        } else {
            y[o] = fminf(fmaxf(x[o] * sc + sh, 0.f), 20.f);
        }
    }, blockIdx.x, gridDim.x, threadIdx.x);
}

__global__ __launch_bounds__(256) void bn1d_apply_kernel(const float* __restrict__ xa, const float* __restrict__ xb,
                                                         const float* __restrict__ mean_invstd,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, size_t n, int F,
                                                         float* __restrict__ y) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const int col = (int)(i % F);
        float v = xa[i];
        if (xb) v += xb[i];
        const float sc = gamma[col] * mean_invstd[F + col];
        y[i] = (v - mean_invstd[col]) * sc + beta[col];
    }
}

__global__ __launch_bounds__(256) void bn1d_bwd_apply_kernel(const float* __restrict__ xa, const float* __restrict__ xb,
                                                             const float* __restrict__ dy,
                                                             const float* __restrict__ mean_invstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ coef, size_t n, int F,
                                                             float* __restrict__ dx) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const int col = (int)(i % F);
        float v = xa[i];
        if (xb) v += xb[i];
        const float is = mean_invstd[F + col];
        const float xh = (v - mean_invstd[col]) * is;
        dx[i] = gamma[col] * is * (dy[i] - coef[col] - xh * coef[F + col]);
    }
}

// ---- 16-byte forms of the three sequence-flavour kernels (F % 4 == 0, 16-byte aligned rows): the scalar forms above move
// 4 bytes per lane and pay an integer modulo per element -- 0.8 TB/s on (4050 x 800) tensors; these read and write whole
// float4 columns quads.  Per column the rows are summed in exactly the same order, so results are bit-identical.
template <int MODE>
__global__ __launch_bounds__(256) void bn1d_reduce_vec_kernel(const f32x4* __restrict__ xa, const f32x4* __restrict__ xb,
                                                              const f32x4* __restrict__ dy,
                                                              const float* __restrict__ mean_invstd, int rows, int F,
                                                              double* __restrict__ part) {
    const int p = blockIdx.x, tid = threadIdx.x;
    const int F4 = F >> 2;
    const int cq = blockIdx.y * 64 + (tid & 63), ry = tid >> 6;
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = mu;
    if (MODE == 1 && cq < F4) {
        mu = *reinterpret_cast<const f32x4*>(mean_invstd + 4 * cq);
        is = *reinterpret_cast<const f32x4*>(mean_invstd + F + 4 * cq);
    }
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (cq < F4) {
        for (int r = p * 4 + ry; r < rows; r += NPART * 4) {
            const size_t o = (size_t)r * F4 + cq;
            f32x4 v = xa[o];
            if (xb) v += xb[o];
            if (MODE == 0) {
                s0 += v;
                s1 += v * v;
            } else {
                const f32x4 g = dy[o];
                s0 += g;
                s1 += g * (v - mu) * is;
            }
        }
    }
    __shared__ f32x4 sm[4][64][2];
    sm[ry][tid & 63][0] = s0;
    sm[ry][tid & 63][1] = s1;
    __syncthreads();
    if (ry == 0 && cq < F4) {
        const int l = tid & 63;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double d0 = (double)sm[0][l][0][e] + (double)sm[1][l][0][e] + (double)sm[2][l][0][e] + (double)sm[3][l][0][e];
            const double d1 = (double)sm[0][l][1][e] + (double)sm[1][l][1][e] + (double)sm[2][l][1][e] + (double)sm[3][l][1][e];
            part[((size_t)(4 * cq + e) * NPART + p) * 2 + 0] = d0;
            part[((size_t)(4 * cq + e) * NPART + p) * 2 + 1] = d1;
        }
    }
}

__global__ __launch_bounds__(256) void bn1d_apply_vec_kernel(const f32x4* __restrict__ xa, const f32x4* __restrict__ xb,
                                                             const float* __restrict__ mean_invstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, size_t n4, int F4,
                                                             f32x4* __restrict__ y) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const int c = 4 * (int)(i % F4);
        f32x4 v = xa[i];
        if (xb) v += xb[i];
        const f32x4 sc = *reinterpret_cast<const f32x4*>(gamma + c) * *reinterpret_cast<const f32x4*>(mean_invstd + 4 * F4 + c);
        y[i] = (v - *reinterpret_cast<const f32x4*>(mean_invstd + c)) * sc + *reinterpret_cast<const f32x4*>(beta + c);
    }
}

__global__ __launch_bounds__(256) void bn1d_bwd_apply_vec_kernel(const f32x4* __restrict__ xa, const f32x4* __restrict__ xb,
                                                                 const f32x4* __restrict__ dy,
                                                                 const float* __restrict__ mean_invstd,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ coef, size_t n4, int F4,
                                                                 f32x4* __restrict__ dx) {
    const size_t stride = (size_t)gridDim.x * 256;
    const int F = 4 * F4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const int c = 4 * (int)(i % F4);
        f32x4 v = xa[i];
        if (xb) v += xb[i];
        const f32x4 is = *reinterpret_cast<const f32x4*>(mean_invstd + F + c);
        const f32x4 xh = (v - *reinterpret_cast<const f32x4*>(mean_invstd + c)) * is;
        dx[i] = *reinterpret_cast<const f32x4*>(gamma + c) * is *
                (dy[i] - *reinterpret_cast<const f32x4*>(coef + c) - xh * *reinterpret_cast<const f32x4*>(coef + F + c));
    }
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
inline bool vec4_ok(int F, const void* a, const void* b, const void* c, const void* d) {
    return (F & 3) == 0 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d)) & 15) == 0;
}

inline int ew_blocks(size_t n) {
    size_t b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" size_t ds2_bn_ws_bytes(int C) { return (size_t)C * NPART * 2 * sizeof(double) + (size_t)C * 2 * sizeof(float); }

extern "C" int ds2_bn2d_stats(const float* x, int B, int C, int inner, float eps, float momentum, int use_running,
                              float* running_mean, float* running_var, float* mean_invstd, void* ws, void* stream) {
    DS2_CHECK_ARG(x && mean_invstd && ws && B > 0 && C > 0 && inner > 0);
    DS2_CHECK_ARG(!use_running || (running_mean && running_var));
    hipStream_t st = (hipStream_t)stream;
    double* part = (double*)ws;
    if (!use_running) {
        if (aligned16(x))
            hipLaunchKernelGGL((bn2d_reduce_vec_kernel<0>), dim3(NPART, C), dim3(256), 0, st, x, nullptr, nullptr, nullptr,
                               nullptr, B, C, inner, part);
        else
            hipLaunchKernelGGL((bn2d_reduce_kernel<0>), dim3(NPART, C), dim3(256), 0, st, x, nullptr, nullptr, nullptr,
                               nullptr, B, C, inner, part);
    }
    hipLaunchKernelGGL(bn_finalize_stats_kernel, dim3(C), dim3(64), 0, st, part, C, (double)B * inner, eps, momentum,
                       use_running, running_mean, running_var, mean_invstd);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" int ds2_bn2d_apply_htanh(const float* x, const float* mean_invstd, const float* gamma, const float* beta,
                                    int B, int C, int D, int T, int layout_tbf, float* y, void* stream) {
    DS2_CHECK_ARG(x && mean_invstd && gamma && beta && y && B > 0 && C > 0 && D > 0 && T > 0);
    hipStream_t st = (hipStream_t)stream;
    if (layout_tbf) {
        DS2_CHECK_ARG(B <= 65535);
        dim3 grid(ds2_cdiv(T, 32), ds2_cdiv(C * D, 32), B);
        hipLaunchKernelGGL(bn2d_apply_tbf_kernel, grid, dim3(256), 0, st, x, mean_invstd, gamma, beta, B, C, D, T, y);
    } else {
        const int inner = D * T;
        if (aligned16(x) && aligned16(y)) {
            dim3 grid(min(ds2_cdiv(inner, 1024), 64), B * C);
            hipLaunchKernelGGL(bn2d_apply_vec_kernel, grid, dim3(256), 0, st, x, mean_invstd, gamma, beta, C, inner, y);
        } else {
            dim3 grid(min(ds2_cdiv(inner, 256), 64), B * C);
            hipLaunchKernelGGL(bn2d_apply_kernel, grid, dim3(256), 0, st, x, mean_invstd, gamma, beta, C, inner, y);
        }
    }
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" int ds2_bn2d_htanh_bwd(const float* x, const float* dy, const float* mean_invstd, const float* gamma,
                                  const float* beta, int B, int C, int D, int T, float* dx, float* dgamma,
                                  float* dbeta, void* ws, void* stream) {
    DS2_CHECK_ARG(x && dy && mean_invstd && gamma && beta && dx && dgamma && dbeta && ws);
    DS2_CHECK_ARG(B > 0 && C > 0 && D > 0 && T > 0);
    hipStream_t st = (hipStream_t)stream;
    const int inner = D * T;
    double* part = (double*)ws;
    float* coef = (float*)(part + (size_t)C * NPART * 2);
    const bool vec = aligned16(x) && aligned16(dy) && aligned16(dx);
    if (vec)
        hipLaunchKernelGGL((bn2d_reduce_vec_kernel<1>), dim3(NPART, C), dim3(256), 0, st, x, dy, mean_invstd, gamma, beta,
                           B, C, inner, part);
    else
        hipLaunchKernelGGL((bn2d_reduce_kernel<1>), dim3(NPART, C), dim3(256), 0, st, x, dy, mean_invstd, gamma, beta, B,
                           C, inner, part);
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(C), dim3(64), 0, st, part, C, (double)B * inner, dgamma, dbeta,
                       coef);
    if (vec) {
        dim3 grid(min(ds2_cdiv(inner, 1024), 64), B * C);
        hipLaunchKernelGGL(bn2d_bwd_apply_vec_kernel, grid, dim3(256), 0, st, x, dy, mean_invstd, gamma, beta, coef, C,
                           inner, dx);
    } else {
        dim3 grid(min(ds2_cdiv(inner, 256), 64), B * C);
        hipLaunchKernelGGL(bn2d_bwd_apply_kernel, grid, dim3(256), 0, st, x, dy, mean_invstd, gamma, beta, coef, C, inner,
                           dx);
    }
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" int ds2_bn1d_stats(const float* xa, const float* xb, int rows, int F, float eps, float momentum,
                              int use_running, float* running_mean, float* running_var, float* mean_invstd, void* ws,
                              void* stream) {
    DS2_CHECK_ARG(xa && mean_invstd && ws && rows > 0 && F > 0);
    DS2_CHECK_ARG(!use_running || (running_mean && running_var));
    hipStream_t st = (hipStream_t)stream;
    double* part = (double*)ws;
    if (!use_running) {
        if (vec4_ok(F, xa, xb, nullptr, nullptr))
            hipLaunchKernelGGL((bn1d_reduce_vec_kernel<0>), dim3(NPART, ds2_cdiv(F / 4, 64)), dim3(256), 0, st,
                               (const f32x4*)xa, (const f32x4*)xb, nullptr, nullptr, rows, F, part);
        else
            hipLaunchKernelGGL((bn1d_reduce_kernel<0>), dim3(NPART, ds2_cdiv(F, 64)), dim3(256), 0, st, xa, xb, nullptr,
                               nullptr, rows, F, part);
    }
    hipLaunchKernelGGL(bn_finalize_stats_kernel, dim3(F), dim3(64), 0, st, part, F, (double)rows, eps, momentum,
                       use_running, running_mean, running_var, mean_invstd);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" int ds2_bn1d_apply(const float* xa, const float* xb, const float* mean_invstd, const float* gamma,
                              const float* beta, int rows, int F, float* y, void* stream) {
    DS2_CHECK_ARG(xa && mean_invstd && gamma && beta && y && rows > 0 && F > 0);
    const size_t n = (size_t)rows * F;
    if (vec4_ok(F, xa, xb, y, gamma) && vec4_ok(F, beta, mean_invstd, nullptr, nullptr))
        hipLaunchKernelGGL(bn1d_apply_vec_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream,
                           (const f32x4*)xa, (const f32x4*)xb, mean_invstd, gamma, beta, n / 4, F / 4, (f32x4*)y);
    else
        hipLaunchKernelGGL(bn1d_apply_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, xa, xb,
                           mean_invstd, gamma, beta, n, F, y);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" int ds2_bn1d_bwd(const float* xa, const float* xb, const float* dy, const float* mean_invstd,
                            const float* gamma, int rows, int F, float* dx, float* dgamma, float* dbeta, void* ws,
                            void* stream) {
    DS2_CHECK_ARG(xa && dy && mean_invstd && gamma && dx && dgamma && dbeta && ws && rows > 0 && F > 0);
    hipStream_t st = (hipStream_t)stream;
    double* part = (double*)ws;
    float* coef = (float*)(part + (size_t)F * NPART * 2);
    const bool vec = vec4_ok(F, xa, xb, dy, dx) && vec4_ok(F, gamma, mean_invstd, nullptr, nullptr);
    if (vec)
        hipLaunchKernelGGL((bn1d_reduce_vec_kernel<1>), dim3(NPART, ds2_cdiv(F / 4, 64)), dim3(256), 0, st,
                           (const f32x4*)xa, (const f32x4*)xb, (const f32x4*)dy, mean_invstd, rows, F, part);
    else
        hipLaunchKernelGGL((bn1d_reduce_kernel<1>), dim3(NPART, ds2_cdiv(F, 64)), dim3(256), 0, st, xa, xb, dy,
                           mean_invstd, rows, F, part);
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(F), dim3(64), 0, st, part, F, (double)rows, dgamma, dbeta, coef);
    const size_t n = (size_t)rows * F;
    if (vec)
        hipLaunchKernelGGL(bn1d_bwd_apply_vec_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, st, (const f32x4*)xa,
                           (const f32x4*)xb, (const f32x4*)dy, mean_invstd, gamma, coef, n / 4, F / 4, (f32x4*)dx);
    else
        hipLaunchKernelGGL(bn1d_bwd_apply_kernel, dim3(ew_blocks(n)), dim3(256), 0, st, xa, xb, dy, mean_invstd, gamma,
                           coef, n, F, dx);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

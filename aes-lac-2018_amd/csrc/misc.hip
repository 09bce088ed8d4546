// Small HBM-bound helpers: transposes, row softmax / argmax, greedy collapse, add.
#include "ds2_common.h"

namespace {

// out[c][r] = in[r][c] per batch; 32x32 tiles through LDS (padded), coalesced both sides.
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int rows, int cols,
                                                        float* __restrict__ out) {
    __shared__ float tile[32][33];
    const size_t boff = (size_t)blockIdx.z * rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int r = r0 + ty + i, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + i][tx] = in[boff + (size_t)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int c = c0 + ty + i, r = r0 + tx;
        if (r < rows && c < cols) out[boff + (size_t)c * rows + r] = tile[tx][ty + i];
    }
}

// the same for up to 8 separately placed inputs of `batch` matrices each, ONE launch: out is [input][batch][cols][rows]
struct TransposeGroup { const float* in[8]; };
__global__ __launch_bounds__(256) void transpose_group_kernel(TransposeGroup g, int batch, int rows, int cols,
                                                              float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int which = blockIdx.z / batch, b = blockIdx.z % batch;
    const float* __restrict__ in = g.in[which] + (size_t)b * rows * cols;
    float* __restrict__ o = out + (size_t)blockIdx.z * rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int r = r0 + ty + i, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + i][tx] = in[(size_t)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int c = c0 + ty + i, r = r0 + tx;
        if (r < rows && c < cols) o[(size_t)c * rows + r] = tile[tx][ty + i];
    }
}

// one wave per row, A <= 64 * 4
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, int rows, int A,
                                                           float* __restrict__ y) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* p = x + (size_t)row * A;
    float v[4];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane + 64 * i;
        v[i] = k < A ? p[k] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = (lane + 64 * i < A) ? expf(v[i] - mx) : 0.f;
        sum += v[i];
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane + 64 * i;
        if (k < A) y[(size_t)row * A + k] = v[i] * inv;
    }
}

// argmax with ties -> lowest index (torch.max semantics); one wave per row
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int rows, int A,
                                                          int32_t* __restrict__ idx) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* p = x + (size_t)row * A;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int k = lane; k < A; k += 64) {
        const float v = p[k];
        if (v > best || (v == best && k < bi)) {
            best = v;
            bi = k;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        const bool take = (ov > best) || (ov == best && oi < bi);
        if (take) {
            best = ov;
            bi = oi;
        }
    }
    if (lane == 0) idx[row] = bi;
}

// One wave per utterance: keep frame t iff t < size, c != blank and (t == 0 or c != best[t-1]).
__global__ __launch_bounds__(64) void greedy_collapse_kernel(const int32_t* __restrict__ best,
                                                             const int32_t* __restrict__ sizes, int T, int blank,
                                                             int32_t* __restrict__ out_ids,
                                                             int32_t* __restrict__ out_off,
                                                             int32_t* __restrict__ out_lens) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int32_t* row = best + (size_t)b * T;
    const int n = min(sizes[b], T);
    int count = 0;
    for (int t0 = 0; t0 < n; t0 += 64) {
        const int t = t0 + lane;
        bool keep = false;
        int c = blank;
        if (t < n) {
            c = row[t];
            keep = (c != blank) && (t == 0 || c != row[t - 1]);
        }
        const unsigned long long mask = __ballot(keep);
        const int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
        if (keep) {
            out_ids[(size_t)b * T + pos] = c;
            out_off[(size_t)b * T + pos] = t;
        }
        count += __popcll(mask);
    }
    if (lane == 0) out_lens[b] = count;
}

__global__ __launch_bounds__(256) void add2_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                   size_t n4, size_t n, float* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 x = reinterpret_cast<const f32x4*>(a)[i];
        const f32x4 y = reinterpret_cast<const f32x4*>(b)[i];
        reinterpret_cast<f32x4*>(out)[i] = x + y;
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) out[i] = a[i] + b[i];
}

}  // namespace

extern "C" int ds2_transpose2d(const float* in, int rows, int cols, float* out, void* stream) {
    DS2_CHECK_ARG(in && out && rows > 0 && cols > 0);
    dim3 grid(ds2_cdiv(cols, 32), ds2_cdiv(rows, 32), 1);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, rows, cols, out);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" int ds2_transpose2d_group(int count, const float* const* in_host, int batch, int rows, int cols, float* out,
                                     void* stream) {
    DS2_CHECK_ARG(in_host && out && count >= 1 && count <= 8 && batch >= 1 && rows > 0 && cols > 0 && count * batch <= 65535);
    TransposeGroup g;
    for (int i = 0; i < 8; ++i) g.in[i] = in_host[i < count ? i : 0];
    for (int i = 0; i < count; ++i) DS2_CHECK_ARG(in_host[i] != nullptr);
    dim3 grid(ds2_cdiv(cols, 32), ds2_cdiv(rows, 32), count * batch);
    hipLaunchKernelGGL(transpose_group_kernel, grid, dim3(256), 0, (hipStream_t)stream, g, batch, rows, cols, out);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" int ds2_transpose_btf_to_bft(const float* x, int B, int T, int F, float* x_t, void* stream) {
    DS2_CHECK_ARG(x && x_t && B > 0 && T > 0 && F > 0 && B <= 65535);
    dim3 grid(ds2_cdiv(F, 32), ds2_cdiv(T, 32), B);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, T, F, x_t);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" int ds2_softmax_rows(const float* x, int rows, int A, float* y, void* stream) {
    DS2_CHECK_ARG(x && y && rows > 0 && A > 0 && A <= 256);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(ds2_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, A,
                       y);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" int ds2_argmax_rows(const float* x, int rows, int A, int32_t* idx, void* stream) {
    DS2_CHECK_ARG(x && idx && rows > 0 && A > 0);
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(ds2_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, A,
                       idx);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" int ds2_greedy_collapse(const int32_t* best, const int32_t* sizes, int B, int T, int blank,
                                   int32_t* out_ids, int32_t* out_offsets, int32_t* out_lens, void* stream) {
    DS2_CHECK_ARG(best && sizes && out_ids && out_offsets && out_lens && B > 0 && T > 0);
    hipLaunchKernelGGL(greedy_collapse_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, best, sizes, T, blank,
                       out_ids, out_offsets, out_lens);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" int ds2_add2(const float* a, const float* b, size_t n, float* out, void* stream) {
    DS2_CHECK_ARG(a && b && out && n > 0);
    const bool al = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0;
    const size_t n4 = al ? n / 4 : 0;
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? ((n4 + 255) / 256 > 0 ? (n4 + 255) / 256 : 1) : 2048);
    hipLaunchKernelGGL(add2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, n4, n, out);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" int ds2_stream_create(int priority, void** stream_out) {
    DS2_CHECK_ARG(stream_out);
    int least = 0, greatest = 0;
    DS2_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    if (priority > least) priority = least;        // numerically larger = lower priority
    if (priority < greatest) priority = greatest;
    hipStream_t st = nullptr;
    DS2_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, priority));
    *stream_out = (void*)st;
    return DS2_OK;
}

extern "C" int ds2_stream_destroy(void* stream) {
    DS2_CHECK_ARG(stream);
    DS2_HIP(hipStreamDestroy((hipStream_t)stream));
    return DS2_OK;
}

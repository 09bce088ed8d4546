// TEST INFRASTRUCTURE (tests/test_coresidency_gpu.py): a stand-in for an RCCL channel kernel.  RCCL's all-reduce of the flat
// gradient runs as a few dozen long-lived workgroups (NCCL_MAX_NCHANNELS = 32 in bench.py / train.py) that stream HBM while
// the persistent recurrence kernels -- which need ~204 whole CUs co-resident -- and the side stream's weight-gradient GEMMs
// share the chip.  No multi-GPU node is available to the builder, so this kernel occupies `wgs` workgroups of 256 threads
// for about `iters` passes over a buffer (read + write, 16 bytes per lane: HBM streaming, ~1 ms for the defaults the test
// uses); the test launches it back to back on a third stream for the whole duration of a training step.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void co_resident_stream_kernel(f32x4* __restrict__ buf, size_t n16, int iters) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (int it = 0; it < iters; ++it)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
            f32x4 v = buf[i];
            v = v * 1.0000001f + 1e-9f;
            buf[i] = v;
        }
}

extern "C" int co_resident_stream(void* buf, size_t bytes, int wgs, int iters, void* stream) {
    if (!buf || bytes < 16 || wgs <= 0 || iters <= 0) return -1;
    hipLaunchKernelGGL(co_resident_stream_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, (f32x4*)buf, bytes / 16, iters);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

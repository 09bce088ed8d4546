// Probe: how long until a wave's stores are acknowledged (s_waitcnt vmcnt(0) returns), per store flavour, with one
// 512-thread workgroup on every CU doing the same thing.  Prints shader-clock cycles (s_memtime), median over steps.
//   hipcc --offload-arch=gfx950 -O2 -w tools/store_ack_probe.hip -o tools/store_ack_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

// MODE 0: plain dword stores to a small ring (L2 resident, rewritten every step)   1: the same, sc1 (write-through)
//      2: plain dword stores to fresh lines every step (streaming, like the saved activations)   3: 2 + ring plain   4: loads only (HBM fresh)
template <int MODE>
__global__ __launch_bounds__(512) void probe(float* ring, float* stream, long long* out, int steps) {
    const int tid = threadIdx.x;
    float* myring = ring + blockIdx.x * 1024;
    float acc = 0.f;
    for (int s = 0; s < steps; ++s) {
        __syncthreads();
        const long long t0 = __builtin_amdgcn_s_memtime();
        float* fresh = stream + ((size_t)s * gridDim.x + blockIdx.x) * 512;
        if (tid < 112) {
            if (MODE == 0 || MODE == 3) __hip_atomic_store(&myring[(s & 3) * 128 + tid], (float)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 1) __hip_atomic_store(&myring[(s & 3) * 128 + tid], (float)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (MODE == 2 || MODE == 3) {
                fresh[tid] = (float)s;
                fresh[128 + tid] = (float)s;
                fresh[256 + tid] = (float)s;
                fresh[384 + tid] = (float)s;
            }
            if (MODE == 4) acc += fresh[tid] + fresh[128 + tid] + fresh[256 + tid] + fresh[384 + tid];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t1 = __builtin_amdgcn_s_memtime();
        if (tid == 0 && blockIdx.x == 7) out[s] = t1 - t0;
        for (int i = 0; i < 40; ++i) __builtin_amdgcn_s_sleep(8);   // ~1 us between steps
    }
    if (acc == 1234.5f) ring[0] = acc;
}

template <int MODE>
void run(const char* name, float* ring, float* stream, long long* out) {
    const int steps = 300;
    hipMemset(out, 0, steps * 8);
    probe<MODE><<<256, 512>>>(ring, stream, out, steps);
    hipDeviceSynchronize();
    std::vector<long long> h(steps);
    hipMemcpy(h.data(), out, steps * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin() + 20, h.end());
    printf("%-60s median %lld cycles, p90 %lld\n", name, h[20 + (steps - 20) / 2], h[20 + (steps - 20) * 9 / 10]);
}

int main() {
    float *ring, *stream;
    long long* out;
    hipMalloc(&ring, 256 * 1024 * 4);
    hipMalloc(&stream, (size_t)300 * 256 * 512 * 4);
    hipMalloc(&out, 300 * 8);
    hipMemset(stream, 0, (size_t)300 * 256 * 512 * 4);
    run<0>("plain stores, small ring (L2 resident)", ring, stream, out);
    run<1>("sc1 stores, small ring (write-through)", ring, stream, out);
    run<2>("4 plain stores to fresh lines (streaming)", ring, stream, out);
    run<3>("ring plain + 4 streaming", ring, stream, out);
    run<4>("4 plain loads of fresh lines (HBM / Infinity Cache)", ring, stream, out);
    run<0>("plain stores, small ring (L2 resident)", ring, stream, out);
    return 0;
}

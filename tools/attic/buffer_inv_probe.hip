#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void inv_probe(long long* out, float* buf, int n, int mode) {
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if (mode & 1) asm volatile("buffer_inv sc1" ::: "memory");
        if (mode & 2) { acc += buf[(size_t)((blockIdx.x * 131 + i * 17) & 4095) * 64 + lane]; }
        if (mode & 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 12345.f) buf[0] = acc;
}
int main() {
    long long* out; float* buf;
    hipMalloc(&out, 1024 * 8); hipMalloc(&buf, 4096 * 64 * 4 * 2);
    hipMemset(buf, 0, 4096 * 64 * 4 * 2);
    long long h[1024];
    for (int mode : {1, 5, 2, 6, 7}) for (int wgs : {1, 240}) {
        hipLaunchKernelGGL(inv_probe, dim3(wgs), dim3(512), 0, 0, out, buf, 1000, mode);
        hipLaunchKernelGGL(inv_probe, dim3(wgs), dim3(512), 0, 0, out, buf, 1000, mode);
        hipDeviceSynchronize();
        hipMemcpy(h, out, wgs * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < wgs; ++i) s += h[i];
        printf("mode %d (1=inv 2=load 4=wait) wgs %3d: %.1f ticks per iteration (s_memtime, 100 MHz? /clk)\n", mode, wgs, s / wgs / 1000.0);
    }
    return 0;
}

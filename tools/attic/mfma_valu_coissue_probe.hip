// Probe: do the matrix pipe (v_mfma_f32_4x4x1_16b_f32) and the vector pipe (v_fma_f32) of a SIMD run side by side when they
// are fed by DIFFERENT waves of one 512-thread workgroup (waves w and w + 4 share SIMD w % 4)?  One workgroup per CU.
//   mode 0: all 8 waves MFMA      mode 1: all 8 waves v_fma      mode 2: waves 0-3 MFMA, waves 4-7 v_fma (same per-wave counts)
//   mode 3: only waves 0-3 MFMA (4-7 idle)       mode 4: only waves 4-7 v_fma (0-3 idle)
//   hipcc --offload-arch=gfx950 -O2 -w tools/mfma_valu_coissue_probe.hip -o tools/mfma_valu_coissue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void probe(float* out, int iters, int mode, long long* cyc) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || ((mode == 2 || mode == 3) && wave < 4);
    const bool do_valu = mode == 1 || ((mode == 2 || mode == 4) && wave >= 4);
    f32x4 acc[12];
    float v[16];
    for (int i = 0; i < 12; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 16; ++i) v[i] = (float)threadIdx.x * 1e-3f + i;
    const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);   // 12 per iteration
        }
    }
    if (do_valu) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], a, b);   // 48 v_fma per iteration (= 12 MFMA x 8 cyc / 2 cyc)
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const long long t2 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 3 && (threadIdx.x & 63) == 0) {
        cyc[wave] = t1 - t0;
        if (wave == 0) cyc[8] = t2 - t0;
    }
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 16 * 8);
    const int iters = 2000;
    const char* names[] = {"all 8 waves MFMA 4x4x1", "all 8 waves v_fma", "waves 0-3 MFMA + waves 4-7 v_fma", "waves 0-3 MFMA only",
                           "waves 4-7 v_fma only"};
    for (int mode = 0; mode < 5; ++mode) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        probe<<<256, 512>>>(out, iters, mode, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        probe<<<256, 512>>>(out, iters, mode, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        long long h[9];
        hipMemcpy(h, cyc, 9 * 8, hipMemcpyDeviceToHost);
        printf("%-36s %7.3f ms  workgroup %lld cycles; per wave:", names[mode], ms, h[8]);
        for (int w = 0; w < 8; ++w) printf(" %lld", h[w]);
        printf("   (12 MFMA = 96 pipe cycles, 48 v_fma = 96 issue cycles per iteration; x%d)\n", iters);
    }
    return 0;
}

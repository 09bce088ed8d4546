// Probe the lane/register layout and issue rate of v_mfma_f32_4x4x1_16b_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

__global__ void rate(float* out, int iters) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float a = threadIdx.x * 0.001f, b = 1.0f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
    }
    long long t1 = clock64();
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / (4.0f * iters);
}

__global__ void rate12(float* out, int iters) {
    f32x4 c[12];
    for (int i = 0; i < 12; ++i) c[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 1.0f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 12; ++j) c[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[j], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int j = 0; j < 12; ++j) s += c[j][j & 3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / (12.0f * iters);
}

__global__ void rate16(float* out, int iters) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0;
    float a = threadIdx.x * 0.001f, b = 1.0f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
    }
    long long t1 = clock64();
    out[threadIdx.x] = c0[0] + c1[1];
    if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / (2.0f * iters);
}

int main() {
    float *a, *b, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024 + 64);
    std::vector<float> ha(64), hb(64), hd(256);
    // A lane l = 1 + l (identifies (block,row)), B lane l = 100 * (1 + l): D = A*B reveals which lanes pair up
    for (int l = 0; l < 64; ++l) { ha[l] = 1 + l; hb[l] = 100.0f * (1 + l); }
    hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(a, b, d);
    hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int r = 0; r < 4; ++r) {
            // D = (1+la) * 100 * (1+lb) -> recover la, lb
            float v = hd[l * 4 + r] / 100.0f;
            int found = 0;
            for (int la = 0; la < 64 && !found; ++la)
                for (int lb = 0; lb < 64; ++lb)
                    if (v == (float)(1 + la) * (1 + lb) && (la / 4 == lb / 4 || true) && (la / 4 == l / 4) && (lb / 4 == l / 4)) {
                        printf("  r%d=A[l%d]*B[l%d]", r, la, lb); found = 1; break; }
            if (!found) printf("  r%d=%g", r, hd[l * 4 + r]);
        }
        printf("\n");
        if (l == 7) { l = 55; }
    }
    float* o; hipMalloc(&o, 65 * 4);
    std::vector<float> ho(65);
    rate<<<1, 64>>>(o, 100000); hipMemcpy(ho.data(), o, 65 * 4, hipMemcpyDeviceToHost);
    printf("4x4x1_16b: %.2f clock64 ticks per instruction (1 wave)\n", ho[64]);
    rate12<<<1, 64>>>(o, 100000); hipMemcpy(ho.data(), o, 65 * 4, hipMemcpyDeviceToHost);
    printf("4x4x1_16b x12 accumulators: %.2f clock64 ticks per instruction (1 wave)\n", ho[64]);
    rate12<<<1, 128>>>(o, 100000); hipMemcpy(ho.data(), o, 65 * 4, hipMemcpyDeviceToHost);
    printf("4x4x1_16b x12 accumulators: %.2f ticks per instruction per wave (2 waves on the CU)\n", ho[64]);
    rate12<<<1, 512>>>(o, 100000); hipMemcpy(ho.data(), o, 65 * 4, hipMemcpyDeviceToHost);
    printf("4x4x1_16b x12 accumulators: %.2f ticks per instruction per wave (8 waves on the CU, 2 per SIMD)\n", ho[64]);
    rate16<<<1, 512>>>(o, 100000); hipMemcpy(ho.data(), o, 65 * 4, hipMemcpyDeviceToHost);
    printf("16x16x4  : %.2f ticks per instruction per wave (8 waves, 2 per SIMD)\n", ho[64]);
    rate16<<<1, 64>>>(o, 100000); hipMemcpy(ho.data(), o, 65 * 4, hipMemcpyDeviceToHost);
    printf("16x16x4  : %.2f clock64 ticks per instruction (1 wave)\n", ho[64]);
    return 0;
}

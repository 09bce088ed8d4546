// Probe the A-broadcast controls (CBSZ / ABID) of v_mfma_f32_4x4x1_16b_f32 on gfx950:
// which lane's A value does each block use?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID>
__global__ void probe(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, CBSZ, ABID, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

template <int CBSZ, int ABID>
void run(const float* a, const float* b, float* d) {
    std::vector<float> hd(256);
    probe<CBSZ, ABID><<<1, 64>>>(a, b, d);
    hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
    printf("cbsz=%d abid=%d: A source lane of (lane, reg r): ", CBSZ, ABID);
    // A lane l = prime-ish code (1 + l), B lane l = 1: D[lane][r] = A[src lane]
    for (int l = 0; l < 64; l += 4) printf(" L%02d:[%g %g %g %g]", l, hd[l * 4] - 1, hd[l * 4 + 1] - 1, hd[l * 4 + 2] - 1, hd[l * 4 + 3] - 1);
    printf("\n");
}

void rate_main();
int main() {
    rate_main();
    float *a, *b, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    std::vector<float> ha(64), hb(64);
    for (int l = 0; l < 64; ++l) { ha[l] = 1 + l; hb[l] = 1.0f; }
    hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
    run<0, 0>(a, b, d);
    run<1, 0>(a, b, d); run<1, 1>(a, b, d);
    run<2, 0>(a, b, d); run<2, 1>(a, b, d); run<2, 3>(a, b, d);
    run<3, 5>(a, b, d);
    run<4, 9>(a, b, d);
    return 0;
}

// ---- issue rate of the broadcast forms (run: ./mfma4x4_bcast_probe rate)
template <int CBSZ, int ABID, int NACC>
__global__ void rate(float* out, int iters) {
    f32x4 c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 1.0f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) c[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[j], CBSZ, ABID, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int j = 0; j < NACC; ++j) s += c[j][j & 3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) out[512] = (float)(t1 - t0) / ((float)NACC * iters);
}
template <int CBSZ, int ABID, int NACC>
void run_rate(float* o, int threads) {
    float v;
    rate<CBSZ, ABID, NACC><<<1, threads>>>(o, 100000);
    hipMemcpy(&v, o + 512, 4, hipMemcpyDeviceToHost);
    printf("cbsz=%d abid=%d, %2d accumulators, %d threads: %.2f ticks per instruction per wave\n", CBSZ, ABID, NACC, threads, v);
}
void rate_main() {
    {
        float* o; hipMalloc(&o, 513 * 4);
        run_rate<0, 0, 12>(o, 64); run_rate<0, 0, 12>(o, 512);
        run_rate<2, 1, 12>(o, 64); run_rate<2, 1, 12>(o, 512);
        run_rate<1, 1, 12>(o, 64); run_rate<1, 1, 12>(o, 512);
        run_rate<2, 1, 6>(o, 512); run_rate<2, 1, 3>(o, 512); run_rate<0, 0, 6>(o, 512); run_rate<0, 0, 3>(o, 512);
    }
}

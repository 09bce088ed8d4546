// Log-magnitude STFT frontend on gfx950: one wave per frame, FFT-320 = 5 x 64 in registers.
//
// frame n of a clip covers centre-padded (reflect, 160) samples [160n-160, 160n+160).  Lane l holds
// the 5 windowed samples x[64*n1 + l]; a 5-point DFT over n1 in registers, the W320^(l*k1) twiddle,
// then five 64-point radix-2 DIF FFTs ACROSS the 64 lanes (wave shuffles, no LDS), giving
// X[k1 + 5*k2] with k2 = bitrev6(lane).  Only bins 0..160 are kept: |X| -> log1p -> staged through
// LDS so the 644-byte row leaves as coalesced stores.  Per-frame (sum, sum^2) feed the per-clip
// mean / unbiased std of the second kernel (fp64 combine), which normalises in place.
// Twiddles and the symmetric Hann window are computed once per wave (sincospi) and reused over
// FPW frames.  HBM-bound: 640 B read + 644 B written per frame, then one read+write pass.
#include "ds2_common.h"

namespace {

constexpr int FRAME = 320, HOP = 160, NB = 161;
constexpr int FPW = 4;  // frames per wave

struct cf {
    float re, im;
};
__device__ __forceinline__ cf cmul(cf a, cf b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }

__global__ __launch_bounds__(256) void stft_logmag_kernel(const float* __restrict__ wav,
                                                          const int64_t* __restrict__ offs, int t_max,
                                                          float* __restrict__ out, float* __restrict__ part) {
    __shared__ float stage[4][NB + 3];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t o0 = offs[b];
    const int L = (int)(offs[b + 1] - o0);
    const int nfr = 1 + L / HOP;
    const float* x = wav + o0;

    // per-lane constants: window for the 5 samples, W320^(lane*k1), stage twiddles
    float win[5];
#pragma unroll
    for (int n1 = 0; n1 < 5; ++n1) {
        const int i = 64 * n1 + lane;
        win[n1] = 0.5f - 0.5f * cospif(2.0f * (float)i / (float)(FRAME - 1));
    }
    cf tw320[5];
#pragma unroll
    for (int k1 = 0; k1 < 5; ++k1) {
        float s, c;
        sincospif(-2.0f * (float)((lane * k1) % FRAME) / (float)FRAME, &s, &c);
        tw320[k1] = {c, s};
    }
    cf stw[6];  // stage h = 32 >> st: twiddle W_{2h}^(lane & (h-1))
#pragma unroll
    for (int st = 0; st < 6; ++st) {
        const int h = 32 >> st;
        float s, c;
        sincospif(-(float)(lane & (h - 1)) / (float)h, &s, &c);
        stw[st] = {c, s};
    }
    const int k2 = (int)(__brev((unsigned)lane) >> 26);  // bitrev6

    const float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;
    const float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;

    const int f0 = (blockIdx.x * 4 + wave) * FPW;
    for (int fi = 0; fi < FPW; ++fi) {
        const int n = f0 + fi;
        if (n >= t_max) break;
        float* orow = out + ((size_t)b * t_max + n) * NB;
        if (n >= nfr) {  // collate padding
            for (int k = lane; k < NB; k += 64) orow[k] = 0.f;
            if (lane == 0) {
                part[((size_t)b * t_max + n) * 2] = 0.f;
                part[((size_t)b * t_max + n) * 2 + 1] = 0.f;
            }
            continue;
        }
        float v[5];
#pragma unroll
        for (int n1 = 0; n1 < 5; ++n1) {
            int j = HOP * n - HOP + 64 * n1 + lane;
            if (j < 0) j = -j;
            if (j >= L) j = 2 * (L - 1) - j;
            v[n1] = x[j] * win[n1];
        }
        const float a14 = v[1] + v[4], d14 = v[1] - v[4], a23 = v[2] + v[3], d23 = v[2] - v[3];
        cf y[5];
        y[0] = {v[0] + a14 + a23, 0.f};
        y[1] = {v[0] + c1 * a14 + c2 * a23, -(s1 * d14 + s2 * d23)};
        y[2] = {v[0] + c2 * a14 + c1 * a23, -(s2 * d14 - s1 * d23)};
        y[3] = {y[2].re, -y[2].im};
        y[4] = {y[1].re, -y[1].im};
#pragma unroll
        for (int k1 = 1; k1 < 5; ++k1) y[k1] = cmul(y[k1], tw320[k1]);
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            const int h = 32 >> st;
            const bool upper = (lane & h) != 0;
#pragma unroll
            for (int k1 = 0; k1 < 5; ++k1) {
                const float pr = __shfl_xor(y[k1].re, h, 64), pi = __shfl_xor(y[k1].im, h, 64);
                if (upper) {
                    const cf dlt = {pr - y[k1].re, pi - y[k1].im};
                    y[k1] = cmul(dlt, stw[st]);
                } else {
                    y[k1] = {y[k1].re + pr, y[k1].im + pi};
                }
            }
        }
        float ps = 0.f, pss = 0.f;
#pragma unroll
        for (int k1 = 0; k1 < 5; ++k1) {
            const int k = k1 + 5 * k2;
            if (k < NB) {
                const float m = log1pf(sqrtf(y[k1].re * y[k1].re + y[k1].im * y[k1].im));
                stage[wave][k] = m;
                ps += m;
                pss += m * m;
            }
        }
        ps = wave_sum(ps);
        pss = wave_sum(pss);
        if (lane == 0) {
            part[((size_t)b * t_max + n) * 2] = ps;
            part[((size_t)b * t_max + n) * 2 + 1] = pss;
        }
        // stage[] is private to this wave: a wave-level LDS fence is enough
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        for (int k = lane; k < NB; k += 64) orow[k] = stage[wave][k];
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// grid (chunks, B): every block re-reduces the clip's per-frame partials (<= 1501 pairs) in fp64,
// then normalises its share of the clip's T_in*161 values in place.
__global__ __launch_bounds__(256) void spect_normalize_kernel(const int64_t* __restrict__ offs, int t_max, float eps,
                                                              const float* __restrict__ part,
                                                              float* __restrict__ out) {
    __shared__ double sm[4][2];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int L = (int)(offs[b + 1] - offs[b]);
    const int nfr = min(1 + L / HOP, t_max);
    double s = 0.0, ss = 0.0;
    for (int n = tid; n < nfr; n += 256) {
        s += (double)part[((size_t)b * t_max + n) * 2];
        ss += (double)part[((size_t)b * t_max + n) * 2 + 1];
    }
    s = wave_sum_d(s);
    ss = wave_sum_d(ss);
    if ((tid & 63) == 0) {
        sm[tid >> 6][0] = s;
        sm[tid >> 6][1] = ss;
    }
    __syncthreads();
    s = sm[0][0] + sm[1][0] + sm[2][0] + sm[3][0];
    ss = sm[0][1] + sm[1][1] + sm[2][1] + sm[3][1];
    const double cnt = (double)nfr * NB;
    const double mean = s / cnt;
    double var = (ss - s * s / cnt) / (cnt - 1.0);
    if (var < 0.0) var = 0.0;
    const float fm = (float)mean;
    const float sc = (float)(1.0 / (sqrt(var) + (double)eps));
    float* p = out + (size_t)b * t_max * NB;
    const size_t total = (size_t)nfr * NB;
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < total; i += (size_t)gridDim.x * 256) p[i] = (p[i] - fm) * sc;
}

}  // namespace

extern "C" size_t ds2_spectrogram_ws_bytes(int B, int t_max) { return (size_t)B * t_max * 2 * sizeof(float); }

extern "C" int ds2_spectrogram_fwd(const float* wav, const int64_t* wav_offsets, int B, int t_max, int normalize,
                                   float eps, float* out, void* stats_ws, void* stream) {
    DS2_CHECK_ARG(wav && wav_offsets && out && stats_ws && B > 0 && t_max > 0 && B <= 65535);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(ds2_cdiv(t_max, 4 * FPW), B);
    hipLaunchKernelGGL(stft_logmag_kernel, grid, dim3(256), 0, st, wav, wav_offsets, t_max, out, (float*)stats_ws);
    if (normalize) {
        dim3 g2(min(ds2_cdiv((long)t_max * NB, 256 * 8), 64), B);
        hipLaunchKernelGGL(spect_normalize_kernel, g2, dim3(256), 0, st, wav_offsets, t_max, eps,
                           (const float*)stats_ws, out);
    }
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

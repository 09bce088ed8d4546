// Waveform decode and training-set augmentation on the device (gfx950): 16-bit PCM -> float, WSOLA tempo change,
// gain + 16-bit requantisation.  Replaces the per-clip host work of ToTensor (reference codes/transforms.py:130-224:
// torchaudio.load, and `sox ... tempo T gain G` through a temporary file for every training clip) so that the loader
// only moves int16 bytes: at 350 k frames/s one GPU consumes ~1 h of audio per second.
//
// Everything here is byte / sample shuffling bound by HBM (2 B in, 4 B out per sample) except the WSOLA segment search,
// which is sequential per clip (each segment's search window depends on the previous choice): one workgroup per clip,
// ~0.2 k segments of a 15 s clip, 235 candidate starts x 192 samples of float64 correlation each.  The clips of a
// minibatch run side by side on different CUs.  The arithmetic is specified by oracle/audio.py and matched bit for bit.
#include "ds2_common.h"

namespace {

__global__ __launch_bounds__(256) void pcm16_to_float_kernel(const int16_t* __restrict__ pcm, size_t n, float scale,
                                                             float* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        out[i] = __fmul_rn((float)pcm[i], scale);                        // exact when the scale is a power of two
}

// y = x * gain[b]; q = clip(rint(y * 32768), -32768, 32767); out = q * out_scale   (rint = half to even, as numpy.round)
// x and out may be the same buffer (in place): no __restrict__ on them
__global__ __launch_bounds__(256) void gain_requantize_kernel(const float* x, const int64_t* __restrict__ offsets,
                                                              const float* __restrict__ gain, float out_scale,
                                                              float* out) {
    const int b = blockIdx.y;
    const int64_t lo = offsets[b], n = offsets[b + 1] - lo;
    const float g = gain[b];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float y = __fmul_rn(x[lo + i], g);
        float q = rintf(__fmul_rn(y, 32768.0f));
        q = fminf(fmaxf(q, -32768.0f), 32767.0f);
        out[lo + i] = __fmul_rn(q, out_scale);
    }
}

constexpr int WS_MAX_OVL = 256, WS_MAX_ND = 256;

// One workgroup per clip.  bases[] holds, per segment, the rounded ideal input position (data independent: computed by
// the caller from the clip length and the tempo); the candidate starts are lo..hi around it.
__global__ __launch_bounds__(256) void wsola_kernel(const float* __restrict__ xall, const int64_t* __restrict__ in_off,
                                                    const int64_t* __restrict__ out_off,
                                                    const int32_t* __restrict__ bases,
                                                    const int32_t* __restrict__ base_off, int seg, int ovl, int half,
                                                    float* __restrict__ outall) {
    __shared__ float want[WS_MAX_OVL];
    __shared__ float win[WS_MAX_ND + WS_MAX_OVL];
    __shared__ double wbest[4];
    __shared__ int wbesti[4];
    __shared__ int start_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* x = xall + in_off[b];
    float* out = outall + out_off[b];
    const int n = (int)(in_off[b + 1] - in_off[b]);
    const int nseg = base_off[b + 1] - base_off[b];
    const int32_t* base = bases + base_off[b];
    if (nseg == 0) {                                   // tempo 1, or a clip shorter than one segment + search: a copy
        for (int i = tid; i < n; i += 256) out[i] = x[i];
        return;
    }
    const int hop_out = seg - ovl;
    for (int i = tid; i < seg; i += 256) out[i] = x[i];
    int prev = 0, out_pos = hop_out;
    const double inv_ovl = 1.0 / (double)ovl;
    for (int it = 0; it < nseg; ++it) {
        const int bs = base[it];
        const int lo = max(bs - half, 0), hi = min(bs + half, n - seg);
        const int nd = hi - lo + 1;
        if (tid < ovl) want[tid] = x[prev + hop_out + tid];            // how the previous segment would have gone on
        for (int i = tid; i < nd + ovl - 1; i += 256) win[i] = x[lo + i];
        __syncthreads();
        double c = -INFINITY;
        if (tid < nd) {
            c = 0.0;
            for (int j = 0; j < ovl; ++j) c = fma((double)win[tid + j], (double)want[j], c);   // exact products: order only
        }
        // first maximum over the block: (value, index) with ties -> lowest index
        int ci = tid;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double oc = __shfl_xor(c, o, 64);
            const int oi = __shfl_xor(ci, o, 64);
            if (oc > c || (oc == c && oi < ci)) {
                c = oc;
                ci = oi;
            }
        }
        if ((tid & 63) == 0) {
            wbest[tid >> 6] = c;
            wbesti[tid >> 6] = ci;
        }
        __syncthreads();
        if (tid == 0) {
            double bc = wbest[0];
            int bi = wbesti[0];
            for (int w = 1; w < 4; ++w)
                if (wbest[w] > bc || (wbest[w] == bc && wbesti[w] < bi)) {
                    bc = wbest[w];
                    bi = wbesti[w];
                }
            start_s = lo + bi;
        }
        __syncthreads();
        const int start = start_s;
        if (tid < ovl) {                                                 // cross-fade, every product and the sum rounded
            const float f = (float)((double)tid * inv_ovl);
            out[out_pos + tid] = __fadd_rn(__fmul_rn(want[tid], 1.0f - f), __fmul_rn(x[start + tid], f));
        }
        for (int i = tid; i < seg - ovl; i += 256) out[out_pos + ovl + i] = x[start + ovl + i];
        prev = start;
        out_pos += hop_out;
        __syncthreads();
    }
}

}  // namespace

extern "C" int ds2_pcm16_to_float(const int16_t* pcm, size_t n, float scale, float* out, void* stream) {
    DS2_CHECK_ARG(pcm && out && n > 0 && scale > 0.f);
    int blocks = (int)((n + 1023) / 1024);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pcm16_to_float_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pcm, n, scale, out);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" int ds2_gain_requantize(const float* x, const int64_t* offsets, const float* gain, int B, float out_scale,
                                   float* out, void* stream) {
    DS2_CHECK_ARG(x && offsets && gain && out && B > 0 && B <= 65535 && out_scale > 0.f);
    hipLaunchKernelGGL(gain_requantize_kernel, dim3(64, B), dim3(256), 0, (hipStream_t)stream, x, offsets, gain, out_scale,
                       out);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" int ds2_wsola_tempo(const float* x, const int64_t* in_offsets, const int64_t* out_offsets,
                               const int32_t* bases, const int32_t* base_offsets, int B, int seg, int ovl, int half,
                               float* out, void* stream) {
    DS2_CHECK_ARG(x && in_offsets && out_offsets && bases && base_offsets && out && B > 0);
    DS2_CHECK_ARG(seg >= 4 && ovl >= 1 && ovl <= seg / 2 && half >= 1);
    if (ovl > WS_MAX_OVL || 2 * half + 1 > WS_MAX_ND) {
        ds2_set_error("ds2_wsola_tempo: overlap %d / search %d exceed the kernel's LDS arrays (256 / 256)", ovl,
                      2 * half + 1);
        return DS2_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(wsola_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, in_offsets, out_offsets, bases,
                       base_offsets, seg, ovl, half, out);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

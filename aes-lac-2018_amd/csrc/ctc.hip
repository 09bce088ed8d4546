// CTC negative log-likelihood and its gradient w.r.t. un-normalised activations, gfx950.
//
// Three launches, fp32:
//   K0  row log-sum-exp of acts (T*B rows, one wave per row)                      -> lse
//   K1  grid (B, 2): block (b,0) runs the alpha recursion forward in time, block (b,1) the beta
//       recursion backward in time, concurrently.  The recursions run in the LINEAR domain with
//       a running rescale (Graves' scaled forward-backward): each frame's row is divided by the
//       previous row's sum, and the logs of those sums accumulate in fp64.  That keeps fp32
//       relative accuracy over 746 frames where log-space fp32 loses ~1e-3 (|alpha| ~ 2000), and
//       makes the per-frame work 3 adds + 2 multiplies instead of 3 exp + 1 log.  A row of 2L+1
//       states lives in LDS (double-buffered, one barrier per frame) and is streamed to HBM.
//   K2  grid (T, B): posterior occupancy per symbol w(s) = a(s) b(s) / y(s), normalised per frame
//       by sum_s w(s) (so the arbitrary per-frame scales cancel), grad = softmax - occupancy.
// Only K1 is sequential in T; its critical path is T barriers per utterance.
#include "ds2_common.h"

namespace {

constexpr int CTC_THREADS = 256;
constexpr int MAX_S = 1024;  // 2*L+1 <= 1024

__global__ __launch_bounds__(256) void ctc_lse_kernel(const float* __restrict__ acts, int rows, int A,
                                                      float* __restrict__ lse) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* p = acts + (size_t)row * A;
    float mx = -INFINITY;
    for (int k = lane; k < A; k += 64) mx = fmaxf(mx, p[k]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int k = lane; k < A; k += 64) s += expf(p[k] - mx);
    s = wave_sum(s);
    if (lane == 0) lse[row] = mx + logf(s);
}

__global__ __launch_bounds__(CTC_THREADS) void ctc_alphabeta_kernel(
    const float* __restrict__ acts, const float* __restrict__ lse, const int32_t* __restrict__ labels,
    const int32_t* __restrict__ label_offsets, const int32_t* __restrict__ label_lens,
    const int32_t* __restrict__ act_lens, int T, int B, int A, int smax, float* __restrict__ alpha,
    float* __restrict__ beta, float* __restrict__ ll_out, float* __restrict__ costs) {
    __shared__ int ext[MAX_S];
    __shared__ float rowbuf[2][MAX_S + 4];
    __shared__ float part[2][4];

    const int b = blockIdx.x, dirn = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int L = label_lens[b], tl = min(act_lens[b], T);
    const int S = 2 * L + 1;
    const int32_t* lab = labels + label_offsets[b];
    for (int s = tid; s < S; s += CTC_THREADS) ext[s] = (s & 1) ? lab[s >> 1] : 0;
    __syncthreads();
    if (tl <= 0) {
        if (dirn == 0 && tid == 0) {
            const float c = (L == 0) ? 0.f : INFINITY;
            costs[b] = c;
            ll_out[b] = -c;
        }
        return;
    }
    float* out = (dirn == 0 ? alpha : beta) + (size_t)b * T * smax;
    constexpr int NS = MAX_S / CTC_THREADS;
    int sym[NS];
    bool skip[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int s = tid + i * CTC_THREADS;
        sym[i] = s < S ? ext[s] : 0;
        if (dirn == 0)
            skip[i] = (s < S) && (s >= 2) && (ext[s] != 0) && (ext[s] != ext[s - 2]);
        else
            skip[i] = (s + 2 < S) && (ext[s] != 0) && (ext[s] != ext[s + 2]);
    }
    // rowbuf[.][s + 1] holds state s; cells 0 and S+1.. are zero guards
    const int tstart = dirn == 0 ? 0 : tl - 1;
    const int tstep = dirn == 0 ? 1 : -1;
    for (int i = tid; i < MAX_S + 4; i += CTC_THREADS) {
        rowbuf[0][i] = 0.f;
        rowbuf[1][i] = 0.f;
    }
    __syncthreads();
    {
        const float* arow = acts + ((size_t)tstart * B + b) * A;
        const float l0 = lse[(size_t)tstart * B + b];
        float ps = 0.f;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int s = tid + i * CTC_THREADS;
            if (s < S) {
                const bool on = dirn == 0 ? (s <= 1) : (s >= S - 2);
                const float v = on ? expf(arow[sym[i]] - l0) : 0.f;
                rowbuf[0][s + 1] = v;
                out[(size_t)tstart * smax + s] = v;
                ps += v;
            }
        }
        ps = wave_sum(ps);
        if (lane == 0) part[0][wave] = ps;
    }
    __syncthreads();
    int cur = 0;
    int t = tstart + tstep;
    double logc = 0.0;
    bool dead = false;
    float yv[NS];
    if (tl > 1) {
        const float* arow = acts + ((size_t)t * B + b) * A;
        const float l0 = lse[(size_t)t * B + b];
#pragma unroll
        for (int i = 0; i < NS; ++i) yv[i] = (tid + i * CTC_THREADS < S) ? expf(arow[sym[i]] - l0) : 0.f;
    }
    for (int step = 1; step < tl; ++step) {
        const float* prev = rowbuf[cur];
        float* nxt = rowbuf[cur ^ 1];
        const float sigma = part[cur][0] + part[cur][1] + part[cur][2] + part[cur][3];
        float inv = 0.f;
        if (sigma > 0.f) {
            inv = 1.f / sigma;
            logc += (double)logf(sigma);
        } else {
            dead = true;
        }
        float val[NS];
        float ps = 0.f;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int s = tid + i * CTC_THREADS;
            val[i] = 0.f;
            if (s < S) {
                float acc = prev[s + 1];
                if (dirn == 0) {
                    acc += prev[s];
                    if (skip[i]) acc += prev[s - 1];
                } else {
                    acc += prev[s + 2];
                    if (skip[i]) acc += prev[s + 3];
                }
                val[i] = acc * inv * yv[i];
                ps += val[i];
            }
        }
        const int tn = t + tstep;
        float yn[NS];
        if (step + 1 < tl) {  // prefetch next frame's emissions before the barrier
            const float* arow = acts + ((size_t)tn * B + b) * A;
            const float l0 = lse[(size_t)tn * B + b];
#pragma unroll
            for (int i = 0; i < NS; ++i) yn[i] = (tid + i * CTC_THREADS < S) ? expf(arow[sym[i]] - l0) : 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < NS; ++i) yn[i] = 0.f;
        }
        ps = wave_sum(ps);
        if (lane == 0) part[cur ^ 1][wave] = ps;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int s = tid + i * CTC_THREADS;
            if (s < S) {
                nxt[s + 1] = val[i];
                out[(size_t)t * smax + s] = val[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NS; ++i) yv[i] = yn[i];
        cur ^= 1;
        t = tn;
    }
    if (dirn == 0 && tid == 0) {
        const float* last = rowbuf[cur];
        const float tail = last[S] + (S > 1 ? last[S - 1] : 0.f);  // states S-1 and S-2
        float ll = -INFINITY;
        if (!dead && tail > 0.f) ll = (float)(logc + (double)logf(tail));
        ll_out[b] = ll;
        costs[b] = -ll;
    }
}

__global__ __launch_bounds__(128) void ctc_grad_kernel(const float* __restrict__ acts, const float* __restrict__ lse,
                                                       const int32_t* __restrict__ labels,
                                                       const int32_t* __restrict__ label_offsets,
                                                       const int32_t* __restrict__ label_lens,
                                                       const int32_t* __restrict__ act_lens, int T, int B, int A,
                                                       int smax, const float* __restrict__ alpha,
                                                       const float* __restrict__ beta,
                                                       const float* __restrict__ ll_in, float* __restrict__ grad) {
    __shared__ float occ[256];
    __shared__ float zsum[2];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    float* grow = grad + ((size_t)t * B + b) * A;
    const int tl = min(act_lens[b], T);
    const float ll = ll_in[b];
    if (t >= tl || ll == -INFINITY) {
        for (int k = tid; k < A; k += 128) grow[k] = 0.f;
        return;
    }
    for (int k = tid; k < A; k += 128) occ[k] = 0.f;
    __syncthreads();
    const int L = label_lens[b], S = 2 * L + 1;
    const int32_t* lab = labels + label_offsets[b];
    const float* arow = acts + ((size_t)t * B + b) * A;
    const float l0 = lse[(size_t)t * B + b];
    const float* al = alpha + ((size_t)b * T + t) * smax;
    const float* be = beta + ((size_t)b * T + t) * smax;
    const float y_blank = expf(arow[0] - l0);
    const float inv_yb = y_blank > 0.f ? 1.f / y_blank : 0.f;
    float blank_sum = 0.f, z = 0.f;
    for (int s = tid; s < S; s += 128) {
        const float ab = al[s] * be[s];
        if (ab == 0.f) continue;
        if (s & 1) {
            const int k = lab[s >> 1];
            const float y = expf(arow[k] - l0);
            const float w = y > 0.f ? ab / y : 0.f;
            atomicAdd(&occ[k], w);
            z += w;
        } else {
            blank_sum += ab * inv_yb;
        }
    }
    z = wave_sum(z + blank_sum);
    blank_sum = wave_sum(blank_sum);
    if ((tid & 63) == 0) {
        atomicAdd(&occ[0], blank_sum);
        zsum[tid >> 6] = z;
    }
    __syncthreads();
    const float ztot = zsum[0] + zsum[1];
    const float invz = ztot > 0.f ? 1.f / ztot : 0.f;
    for (int k = tid; k < A; k += 128) grow[k] = expf(arow[k] - l0) - occ[k] * invz;
}

}  // namespace

extern "C" size_t ds2_ctc_ws_bytes(int T, int B, int A, int max_label_len) {
    (void)A;
    const size_t smax = 2 * (size_t)max_label_len + 1;
    return sizeof(float) * ((size_t)T * B + 2 * (size_t)B * T * smax + (size_t)B) + 64;
}

extern "C" int ds2_ctc_loss_grad(const float* acts, const int32_t* labels, const int32_t* label_offsets,
                                 const int32_t* label_lens, const int32_t* act_lens, int T, int B, int A,
                                 int max_label_len, float* costs, float* grad, void* ws, void* stream) {
    DS2_CHECK_ARG(acts && labels && label_offsets && label_lens && act_lens && costs && grad && ws);
    DS2_CHECK_ARG(T > 0 && B > 0 && A > 1 && A <= 256 && max_label_len >= 0);
    const int smax = 2 * max_label_len + 1;
    DS2_CHECK_ARG(smax <= MAX_S);
    DS2_CHECK_ARG(B <= 65535);
    hipStream_t st = (hipStream_t)stream;
    float* lse = (float*)ws;
    float* alpha = lse + (size_t)T * B;
    float* beta = alpha + (size_t)B * T * smax;
    float* ll = beta + (size_t)B * T * smax;
    hipLaunchKernelGGL(ctc_lse_kernel, dim3(ds2_cdiv((long)T * B, 4)), dim3(256), 0, st, acts, T * B, A, lse);
    hipLaunchKernelGGL(ctc_alphabeta_kernel, dim3(B, 2), dim3(CTC_THREADS), 0, st, acts, lse, labels, label_offsets,
                       label_lens, act_lens, T, B, A, smax, alpha, beta, ll, costs);
    hipLaunchKernelGGL(ctc_grad_kernel, dim3(T, B), dim3(128), 0, st, acts, lse, labels, label_offsets, label_lens,
                       act_lens, T, B, A, smax, alpha, beta, ll, grad);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

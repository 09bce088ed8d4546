// CTC negative log-likelihood and its gradient w.r.t. un-normalised activations, gfx950.
//
// Three launches:
//   K0  row log-sum-exp of acts (T*B rows, one wave per row)                      -> lse
//   K1  grid (B, 2): block (b,0) runs the alpha recursion forward in time, block (b,1) the beta
//       recursion backward in time, concurrently, in LOG space.  Values are carried in fp64 but the
//       transcendental part of every log-sum-exp runs in fp32: with m = max(a,b,c),
//           lse3 = m + log(exp(a-m) + exp(b-m) + exp(c-m)),  the log term lies in [0, log 3],
//       so its fp32 rounding is ~1e-7 ABSOLUTE whatever |m| is.  (Plain fp32 log space loses ~1e-3
//       on the posteriors once |alpha| ~ 2000 at T = 746; a rescaled linear-domain recursion
//       underflows fp32 when the posterior mass sits far from the row maximum, e.g. early in training
//       with blank-dominated outputs.)  A row of 2L+1 states lives in LDS (double-buffered, one
//       barrier per frame) and is streamed to HBM for K2.
//   K2  grid (T, B): posterior occupancy per symbol exp(alpha+beta-lp-ll) (LDS float atomics for
//       label states, a wave reduction for the blank states), grad = scale * (softmax - occupancy).
// Only K1 is sequential in T; its critical path is T barriers per utterance.
#include "ds2_common.h"

namespace {

constexpr int CTC_THREADS = 256;
constexpr int MAX_S = 1024;  // 2*L+1 <= 1024
#define NEG_INF_D (-(double)INFINITY)

// exp / log of the bounded part on the hardware exp2 / log2 units (1 ulp): arguments are <= 0 and the sum lies in
// [1, 3], so no range handling is needed; the library expf / logf cost ~100 more instructions per state per frame on
// the recursion's critical path (0.75 -> 0.5 us per frame).
__device__ __forceinline__ float exp_neg(float x) { return __builtin_amdgcn_exp2f(1.4426950408889634f * x); }
__device__ __forceinline__ float log_1to3(float s) { return 0.6931471805599453f * __builtin_amdgcn_logf(s); }

__device__ __forceinline__ double lse2m(double a, double b) {
    const double m = fmax(a, b);
    if (m == NEG_INF_D) return m;
    const float s = exp_neg((float)(a - m)) + exp_neg((float)(b - m));
    return m + (double)log_1to3(s);
}
__device__ __forceinline__ double lse3m(double a, double b, double c) {
    const double m = fmax(fmax(a, b), c);
    if (m == NEG_INF_D) return m;
    const float s = exp_neg((float)(a - m)) + exp_neg((float)(b - m)) + exp_neg((float)(c - m));
    return m + (double)log_1to3(s);
}

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

// NS = states per thread (ceil(smax / 256), 1 .. 4): a template parameter so that a minibatch of short transcripts does not
// carry the dead iterations of the longest supported one on the recursion's critical path
template <int NS>
__global__ __launch_bounds__(CTC_THREADS) void ctc_alphabeta_kernel(
    const float* __restrict__ acts, const float* __restrict__ lse, const int32_t* __restrict__ labels,
    const int32_t* __restrict__ label_offsets, const int32_t* __restrict__ label_lens,
    const int32_t* __restrict__ act_lens, int T, int B, int A, int smax, double* __restrict__ alpha,
    double* __restrict__ beta, double* __restrict__ ll_out, float* __restrict__ costs) {
    __shared__ int ext[MAX_S];
    __shared__ double rowbuf[2][MAX_S + 4];

    const int b = blockIdx.x, dirn = blockIdx.y, tid = threadIdx.x;
    const int L = label_lens[b], tl = min(act_lens[b], T);
    const int S = 2 * L + 1;
    const int32_t* lab = labels + label_offsets[b];
    for (int s = tid; s < S; s += CTC_THREADS) ext[s] = (s & 1) ? lab[s >> 1] : 0;
    for (int i = tid; i < MAX_S + 4; i += CTC_THREADS) {
        rowbuf[0][i] = NEG_INF_D;
        rowbuf[1][i] = NEG_INF_D;
    }
    __syncthreads();
    if (tl <= 0) {
        if (dirn == 0 && tid == 0) {
            const float c = (L == 0) ? 0.f : INFINITY;
            costs[b] = c;
            ll_out[b] = -(double)c;
        }
        return;
    }
    double* out = (dirn == 0 ? alpha : beta) + (size_t)b * T * smax;
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
    // rowbuf[.][s + 1] holds state s; cells 0 and S+1.. stay -inf guards
    const int tstart = dirn == 0 ? 0 : tl - 1;
    const int tstep = dirn == 0 ? 1 : -1;
    {
        const float* arow = acts + ((size_t)tstart * B + b) * A;
        const float l0 = lse[(size_t)tstart * B + b];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int s = tid + i * CTC_THREADS;
            if (s < S) {
                const bool on = dirn == 0 ? (s <= 1) : (s >= S - 2);
                const double v = on ? (double)(arow[sym[i]] - l0) : NEG_INF_D;
                rowbuf[0][s + 1] = v;
                out[(size_t)tstart * smax + s] = v;
            }
        }
    }
    __syncthreads();
    int cur = 0;
    int t = tstart + tstep;
    float lp[NS];
    if (tl > 1) {
        const float* arow = acts + ((size_t)t * B + b) * A;
        const float l0 = lse[(size_t)t * B + b];
#pragma unroll
        for (int i = 0; i < NS; ++i) lp[i] = (tid + i * CTC_THREADS < S) ? arow[sym[i]] - l0 : 0.f;
    }
    for (int step = 1; step < tl; ++step) {
        const double* prev = rowbuf[cur];
        double* nxt = rowbuf[cur ^ 1];
        double val[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int s = tid + i * CTC_THREADS;
            val[i] = NEG_INF_D;
            if (s < S) {
                const double x0 = prev[s + 1];
                double x1, x2 = NEG_INF_D;
                if (dirn == 0) {
                    x1 = prev[s];
                    if (skip[i]) x2 = prev[s - 1];
                } else {
                    x1 = prev[s + 2];
                    if (skip[i]) x2 = prev[s + 3];
                }
                val[i] = lse3m(x0, x1, x2) + (double)lp[i];
            }
        }
        const int tn = t + tstep;
        float lpn[NS];
        if (step + 1 < tl) {  // prefetch next frame's emissions before the barrier
            const float* arow = acts + ((size_t)tn * B + b) * A;
            const float l0 = lse[(size_t)tn * B + b];
#pragma unroll
            for (int i = 0; i < NS; ++i) lpn[i] = (tid + i * CTC_THREADS < S) ? arow[sym[i]] - l0 : 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < NS; ++i) lpn[i] = 0.f;
        }
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
        for (int i = 0; i < NS; ++i) lp[i] = lpn[i];
        cur ^= 1;
        t = tn;
    }
    if (dirn == 0 && tid == 0) {
        const double* last = rowbuf[cur];
        const double ll = lse2m(last[S], S > 1 ? last[S - 1] : NEG_INF_D);  // states S-1 and S-2
        ll_out[b] = ll;
        costs[b] = (float)(-ll);
    }
}

__global__ __launch_bounds__(128) void ctc_grad_kernel(const float* __restrict__ acts, const float* __restrict__ lse,
                                                       const int32_t* __restrict__ labels,
                                                       const int32_t* __restrict__ label_offsets,
                                                       const int32_t* __restrict__ label_lens,
                                                       const int32_t* __restrict__ act_lens, int T, int B, int A,
                                                       int smax, const double* __restrict__ alpha,
                                                       const double* __restrict__ beta,
                                                       const double* __restrict__ ll_in, float grad_scale,
                                                       int zero_batch_if_inf, float* __restrict__ grad) {
    __shared__ float occ[256];
    __shared__ int any_inf;
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    float* grow = grad + ((size_t)t * B + b) * A;
    const int tl = min(act_lens[b], T);
    const double ll = ll_in[b];
    if (zero_batch_if_inf) {   // the training step's rule (codes/engine.py:27-30): an infinite batch loss is replaced by
                               // 0 * loss, so NO utterance of the batch contributes a gradient
        if (tid == 0) any_inf = 0;
        __syncthreads();
        for (int i = tid; i < B; i += 128)
            if (ll_in[i] == NEG_INF_D || ll_in[i] == -NEG_INF_D) any_inf = 1;
        __syncthreads();
    }
    if (t >= tl || ll == NEG_INF_D || (zero_batch_if_inf && any_inf)) {
        for (int k = tid; k < A; k += 128) grow[k] = 0.f;
        return;
    }
    for (int k = tid; k < A; k += 128) occ[k] = 0.f;
    __syncthreads();
    const int L = label_lens[b], S = 2 * L + 1;
    const int32_t* lab = labels + label_offsets[b];
    const float* arow = acts + ((size_t)t * B + b) * A;
    const float l0 = lse[(size_t)t * B + b];
    const double* al = alpha + ((size_t)b * T + t) * smax;
    const double* be = beta + ((size_t)b * T + t) * smax;
    const double base_blank = ll + (double)(arow[0] - l0);
    float blank_sum = 0.f;
    for (int s = tid; s < S; s += 128) {
        const double ab = al[s] + be[s];
        if (ab == NEG_INF_D) continue;
        if (s & 1) {
            const int k = lab[s >> 1];
            atomicAdd(&occ[k], expf((float)(ab - ll - (double)(arow[k] - l0))));
        } else {
            blank_sum += expf((float)(ab - base_blank));
        }
    }
    blank_sum = wave_sum(blank_sum);
    if ((tid & 63) == 0) atomicAdd(&occ[0], blank_sum);
    __syncthreads();
    for (int k = tid; k < A; k += 128) grow[k] = grad_scale * (expf(arow[k] - l0) - occ[k]);
}

}  // namespace

extern "C" size_t ds2_ctc_ws_bytes(int T, int B, int A, int max_label_len) {
    (void)A;
    const size_t smax = 2 * (size_t)max_label_len + 1;
    return sizeof(double) * (2 * (size_t)B * T * smax + (size_t)B) + sizeof(float) * (size_t)T * B + 64;
}

extern "C" int ds2_ctc_loss_grad(const float* acts, const int32_t* labels, const int32_t* label_offsets,
                                 const int32_t* label_lens, const int32_t* act_lens, int T, int B, int A,
                                 int max_label_len, float grad_scale, int zero_batch_if_inf, float* costs,
                                 float* grad, void* ws, void* stream) {
    DS2_CHECK_ARG(acts && labels && label_offsets && label_lens && act_lens && costs && grad && ws);
    DS2_CHECK_ARG(T > 0 && B > 0 && A > 1 && A <= 256 && max_label_len >= 0);
    const int smax = 2 * max_label_len + 1;
    DS2_CHECK_ARG(smax <= MAX_S);
    DS2_CHECK_ARG(B <= 65535);
    hipStream_t st = (hipStream_t)stream;
    double* alpha = (double*)ws;
    double* beta = alpha + (size_t)B * T * smax;
    double* ll = beta + (size_t)B * T * smax;
    float* lse = (float*)(ll + B);
    hipLaunchKernelGGL(ctc_lse_kernel, dim3(ds2_cdiv((long)T * B, 4)), dim3(256), 0, st, acts, T * B, A, lse);
#define DS2_CTC_AB(N_)                                                                                                  \
    hipLaunchKernelGGL(ctc_alphabeta_kernel<N_>, dim3(B, 2), dim3(CTC_THREADS), 0, st, acts, lse, labels, label_offsets,  \
                       label_lens, act_lens, T, B, A, smax, alpha, beta, ll, costs)
    switch (ds2_cdiv(smax, CTC_THREADS)) {
        case 1: DS2_CTC_AB(1); break;
        case 2: DS2_CTC_AB(2); break;
        case 3: DS2_CTC_AB(3); break;
        default: DS2_CTC_AB(4); break;
    }
#undef DS2_CTC_AB
    hipLaunchKernelGGL(ctc_grad_kernel, dim3(T, B), dim3(128), 0, st, acts, lse, labels, label_offsets, label_lens,
                       act_lens, T, B, A, smax, alpha, beta, ll, grad_scale, zero_batch_if_inf, grad);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

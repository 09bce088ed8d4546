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

// One state per thread: NTHR = 256 / 512 / 1024 threads for up to that many states (2 L + 1 of the longest transcript of the
// minibatch), so a minibatch of short transcripts does not pay the barrier of sixteen waves.
//
// Emissions.  State s of frame t needs acts[t][b][ext[s]] - lse[t][b].  Fetched per frame they put a global-memory round
// trip on every step of the recursion (round 1-3: a "prefetch" whose subtraction made the compiler wait for it at once, and a
// vmcnt(0) that also covered the previous frame's alpha / beta store: 0.47 us per frame, 185 us for T = 391).  Now whole
// frames are staged through LDS, CH at a time: the loads of chunk c + 1 are issued before chunk c's first frame and land in
// the other LDS buffer at its last one -- one memory wait per CH frames, and a frame is LDS reads -> log-sum-exp -> LDS write
// -> barrier.
constexpr int EM_FLOATS = 1024;          // per buffer: CH = min(32, 1024 / A) frames (A = 29: 32, A = 43: 23, A = 256: 4)
constexpr int EM_NLD = 4;                // staging loads per thread and chunk (256 threads do the staging)

template <int NTHR>
__global__ __launch_bounds__(NTHR) void ctc_alphabeta_kernel(
    const float* __restrict__ acts, const float* __restrict__ lse, const int32_t* __restrict__ labels,
    const int32_t* __restrict__ label_offsets, const int32_t* __restrict__ label_lens,
    const int32_t* __restrict__ act_lens, int T, int B, int A, int smax, double* __restrict__ alpha,
    double* __restrict__ beta, double* __restrict__ ll_out, float* __restrict__ costs) {
    __shared__ int ext[MAX_S];
    __shared__ double rowbuf[2][MAX_S + 4];
    __shared__ float em[2][EM_FLOATS];

    const int b = blockIdx.x, dirn = blockIdx.y, tid = threadIdx.x;
    const int L = label_lens[b], tl = min(act_lens[b], T);
    const int S = 2 * L + 1;
    const int32_t* lab = labels + label_offsets[b];
    for (int s = tid; s < S; s += NTHR) ext[s] = (s & 1) ? lab[s >> 1] : 0;
    for (int i = tid; i < MAX_S + 4; i += NTHR) {
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
    const int s = tid;
    const bool live = s < S;
    const int sym = live ? ext[s] : 0;
    const bool skip = dirn == 0 ? (live && (s >= 2) && (ext[s] != 0) && (ext[s] != ext[s - 2]))
                                : ((s + 2 < S) && (ext[s] != 0) && (ext[s] != ext[s + 2]));
    // rowbuf[.][s + 1] holds state s; cells 0 and S+1.. stay -inf guards
    const int i1 = dirn == 0 ? s : s + 2, i2 = dirn == 0 ? max(s - 1, 0) : s + 3;     // cells of the two other predecessors
    const int tstart = dirn == 0 ? 0 : tl - 1;
    const int tstep = dirn == 0 ? 1 : -1;
    if (live) {
        const float* arow = acts + ((size_t)tstart * B + b) * A;
        const float l0 = lse[(size_t)tstart * B + b];
        const bool on = dirn == 0 ? (s <= 1) : (s >= S - 2);
        const double v = on ? (double)(arow[sym] - l0) : NEG_INF_D;
        rowbuf[0][s + 1] = v;
        out[(size_t)tstart * smax + s] = v;
    }
    // staging roles: thread tid < 256 carries elements idx = tid + 256 i of a chunk, idx = f A + k (frame f of the chunk, symbol k)
    const int CH = min(32, EM_FLOATS / A), nper = CH * A;
    const int nsteps = tl - 1;                          // recursion steps 1 .. tl - 1; step n works on frame tstart + tstep n
    const int nchunk = (nsteps + CH - 1) / CH;
    int st_f[EM_NLD], st_k[EM_NLD];
#pragma unroll
    for (int i = 0; i < EM_NLD; ++i) {
        const int idx = tid + 256 * i;
        st_f[i] = (tid < 256 && idx < nper) ? idx / A : -1;
        st_k[i] = idx - (idx / A) * A;
    }
    float ra[EM_NLD], rl[EM_NLD];
    auto stage_load = [&](int c) {
#pragma unroll
        for (int i = 0; i < EM_NLD; ++i) {
            ra[i] = rl[i] = 0.f;
            const int n = 1 + c * CH + st_f[i];
            if (st_f[i] >= 0 && n < tl) {
                const size_t row = (size_t)(tstart + tstep * n) * B + b;
                ra[i] = acts[row * A + st_k[i]];
                rl[i] = lse[row];
            }
        }
    };
    auto stage_store = [&](float* dst) {
#pragma unroll
        for (int i = 0; i < EM_NLD; ++i)
            if (st_f[i] >= 0) {
                float a = ra[i];
                asm volatile("" : "+v"(a));      // keeps the subtraction (and with it the wait for the loads) HERE, at the chunk's
                dst[tid + 256 * i] = a - rl[i];  // last frame: hoisted in front of the frame loop it would wait at the chunk's first
            }
    };
    if (nchunk > 0) {
        stage_load(0);
        stage_store(em[0]);
    }
    __syncthreads();
    int cur = 0;
    for (int c = 0; c < nchunk; ++c) {
        const bool more = c + 1 < nchunk;
        if (more) stage_load(c + 1);                    // in flight while this chunk's frames are worked on
        const float* e = em[c & 1] + sym;
        const int fend = min(CH, nsteps - c * CH);
        int t = tstart + tstep * (1 + c * CH);
        for (int f = 0; f < fend; ++f) {
            const double* prev = rowbuf[cur];
            double* nxt = rowbuf[cur ^ 1];
            if (live) {
                // all four LDS reads go out together (the third predecessor is read whether it counts or not)
                const double x0 = prev[s + 1], x1 = prev[i1], x2r = prev[i2];
                const float ev = e[f * A];
                const double val = lse3m(x0, x1, skip ? x2r : NEG_INF_D) + (double)ev;
                nxt[s + 1] = val;
                out[(size_t)t * smax + s] = val;
            }
            if (more && f == fend - 1) stage_store(em[(c + 1) & 1]);    // (its last readers passed chunk c - 1's final barrier)
            __syncthreads();
            cur ^= 1;
            t += tstep;
        }
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
    hipLaunchKernelGGL(ctc_alphabeta_kernel<N_>, dim3(B, 2), dim3(N_), 0, st, acts, lse, labels, label_offsets,          \
                       label_lens, act_lens, T, B, A, smax, alpha, beta, ll, costs)
    if (smax <= 256) DS2_CTC_AB(256);
    else if (smax <= 512) DS2_CTC_AB(512);
    else DS2_CTC_AB(1024);
#undef DS2_CTC_AB
    hipLaunchKernelGGL(ctc_grad_kernel, dim3(T, B), dim3(128), 0, st, acts, lse, labels, label_offsets, label_lens,
                       act_lens, T, B, A, smax, alpha, beta, ll, grad_scale, zero_batch_if_inf, grad);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

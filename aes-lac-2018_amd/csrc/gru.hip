// Bidirectional GRU recurrence for gfx950: one launch per time step, both directions per launch.
//
// Forward step (per direction d, time t): gh = h_{t-1} W_hh[d]^T, gates, h_t.  A workgroup owns a
// slice of JU hidden units of one direction, i.e. 3*JU rows of W_hh (r,z,n rows of those units),
// so the gate non-linearities fuse behind the product and no gh tensor ever reaches HBM.  The
// product runs on v_mfma_f32_16x16x4_f32 with M = the slice's 3*JU (<=16) weight rows, N = 16 batch
// columns per tile and K = H split over the waves of the workgroup; each lane streams 8
// consecutive k of its row straight from L2 into registers (weights are read once per step per
// workgroup -- no LDS staging; the k order inside an MFMA is permuted identically for A and B).
// Partial tiles are reduced across waves through LDS.
//
// Backward step: dh_{t} = d_out_t + dh.z (carried) + dGH_{t'} W_hh[d] with t' the step processed
// just before; rows of M are now UB hidden units (columns of W_hh, read from the transposed copy
// w_hh_t), K = 3H.  Gate derivatives overwrite the saved activations in place:
//   G: (r,z,n) -> (dr_pre, dz_pre, dn_pre);  ghn: W_hn h -> dn_pre * r.
//
// Launch-per-step (instead of a persistent kernel with grid barriers) follows the measured
// prices in MI355X_MICROARCH.md: a dependent kernel boundary is ~1.5 us, a 256-WG grid barrier
// >= 4 us.
#include "ds2_common.h"

namespace {

constexpr int NW = 5;  // waves per workgroup (H = 800 -> 25 k-blocks of 32 -> 5 per wave)

__device__ __forceinline__ void load8(const float* p, bool ok, f32x4& lo, f32x4& hi) {
    if (ok) {
        lo = *reinterpret_cast<const f32x4*>(p);
        hi = *reinterpret_cast<const f32x4*>(p + 4);
    } else {
        lo = f32x4{0.f, 0.f, 0.f, 0.f};
        hi = lo;
    }
}

// ------------------------------------------------------------------------------------------ forward
template <int JU, int NBT>
__global__ __launch_bounds__(NW * 64) void gru_fwd_step_kernel(float* __restrict__ G, float* __restrict__ ghn,
                                                               float* __restrict__ hout,
                                                               const float* __restrict__ w_hh, int T, int B,
                                                               int H, int s) {
    static_assert(3 * JU <= 16, "slice must fit one MFMA M tile");
    static_assert(NBT * 16 * JU <= NW * 64, "one gate element per thread");
    __shared__ float red[NW][NBT][16][17];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dir = blockIdx.y;
    const int j0 = blockIdx.x * JU;
    const int t = dir == 0 ? s : T - 1 - s;
    const bool first = (s == 0);
    const float* hprev = nullptr;
    if (!first) hprev = hout + ((size_t)dir * T + (dir == 0 ? t - 1 : t + 1)) * B * H;

    // --- gate-thread role: prefetch gi and h_prev for (bt, n, jj) before the product
    const int jj = tid % JU, nn = (tid / JU) & 15, gbt = tid / (16 * JU);
    const int gb = gbt * 16 + nn, gj = j0 + jj;
    const bool gate_ok = (gbt < NBT) && (gb < B) && (gj < H);
    float gi_r = 0.f, gi_z = 0.f, gi_n = 0.f, hp = 0.f;
    size_t gbase = 0;
    if (gate_ok) {
        gbase = (((size_t)t * B + gb) * 2 + dir) * 3 * H + gj;
        gi_r = G[gbase];
        gi_z = G[gbase + H];
        gi_n = G[gbase + 2 * H];
        if (!first) hp = hprev[(size_t)gb * H + gj];
    }

    f32x4 acc[NBT];
#pragma unroll
    for (int i = 0; i < NBT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (!first) {
        const int m = lane & 15, q = lane >> 4;
        const int g = m / JU, mj = m % JU;
        const bool a_ok = (m < 3 * JU) && (j0 + mj < H);
        const float* arow = w_hh + ((size_t)dir * 3 * H + (size_t)g * H + j0 + mj) * H;
        const int nkb = (H + 31) >> 5;
        for (int kb = wave; kb < nkb; kb += NW) {
            const int k = kb * 32 + q * 8;
            const bool k_ok = k < H;  // H % 8 == 0, so a chunk is wholly in or out
            f32x4 alo, ahi;
            load8(arow + k, a_ok && k_ok, alo, ahi);
            f32x4 blo[NBT], bhi[NBT];
#pragma unroll
            for (int i = 0; i < NBT; ++i) {
                const int b = i * 16 + m;
                load8(hprev + (size_t)b * H + k, (b < B) && k_ok, blo[i], bhi[i]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < NBT; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(alo[e], blo[i][e], acc[i], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < NBT; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahi[e], bhi[i][e], acc[i], 0, 0, 0);
        }
        // D map of 16x16x4: col = lane & 15 (batch), row = 4 * (lane >> 4) + reg (weight row)
#pragma unroll
        for (int i = 0; i < NBT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][i][4 * q + r][m] = acc[i][r];
    }
    __syncthreads();

    if (gate_ok) {
        float gh_r = 0.f, gh_z = 0.f, gh_n = 0.f;
        if (!first) {
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                gh_r += red[w][gbt][jj][nn];
                gh_z += red[w][gbt][JU + jj][nn];
                gh_n += red[w][gbt][2 * JU + jj][nn];
            }
        }
        const float r = 1.f / (1.f + expf(-(gi_r + gh_r)));
        const float z = 1.f / (1.f + expf(-(gi_z + gh_z)));
        const float n = tanhf(gi_n + r * gh_n);
        const float h = (1.f - z) * n + z * hp;
        G[gbase] = r;
        G[gbase + H] = z;
        G[gbase + 2 * H] = n;
        ghn[(((size_t)t * B + gb) * 2 + dir) * H + gj] = gh_n;
        hout[(((size_t)dir * T + t) * B + gb) * H + gj] = h;
    }
}

// ------------------------------------------------------------------------------------------ backward
template <int UB, int NBT>
__global__ __launch_bounds__(NW * 64) void gru_bwd_step_kernel(float* __restrict__ G, float* __restrict__ ghn,
                                                               const float* __restrict__ hout,
                                                               const float* __restrict__ d_out,
                                                               const float* __restrict__ w_hh_t,
                                                               float* __restrict__ dhz, int T, int B, int H,
                                                               int s) {
    static_assert(UB <= 16, "slice must fit one MFMA M tile");
    static_assert(NBT * 16 * UB <= NW * 64 * 4, "at most 4 gate elements per thread");
    __shared__ float red[NW][NBT][16][17];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dir = blockIdx.y;
    const int j0 = blockIdx.x * UB;
    const int t = dir == 0 ? T - 1 - s : s;
    const int tnext = dir == 0 ? t + 1 : t - 1;  // the step handled by the previous launch
    const bool first = (s == 0);
    const bool has_prev = dir == 0 ? (t > 0) : (t < T - 1);
    const float* hprev = has_prev ? hout + ((size_t)dir * T + (dir == 0 ? t - 1 : t + 1)) * B * H : nullptr;

    f32x4 acc[NBT];
#pragma unroll
    for (int i = 0; i < NBT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (!first) {
        const int m = lane & 15, q = lane >> 4;
        const bool a_ok = (m < UB) && (j0 + m < H);
        const int K = 3 * H;
        const float* arow = w_hh_t + ((size_t)dir * H + j0 + m) * K;
        const int nkb = (K + 31) >> 5;
        for (int kb = wave; kb < nkb; kb += NW) {
            const int k = kb * 32 + q * 8;
            const bool k_ok = k < K;
            f32x4 alo, ahi;
            load8(arow + k, a_ok && k_ok, alo, ahi);
            f32x4 blo[NBT], bhi[NBT];
#pragma unroll
            for (int i = 0; i < NBT; ++i) {
                const int b = i * 16 + m;
                const size_t row = ((size_t)tnext * B + b) * 2 + dir;
                const float* src = (k < 2 * H) ? (G + row * 3 * H + k) : (ghn + row * H + (k - 2 * H));
                load8(src, (b < B) && k_ok, blo[i], bhi[i]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < NBT; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(alo[e], blo[i][e], acc[i], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < NBT; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahi[e], bhi[i][e], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NBT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][i][4 * q + r][m] = acc[i][r];
    }
    // All waves must have finished READING step tnext's dGH before anyone overwrites... (different
    // rows: this launch writes step t only) -- the barrier is for the LDS reduction.
    __syncthreads();

    for (int idx = tid; idx < NBT * 16 * UB; idx += NW * 64) {
        const int mj = idx % UB, nn = (idx / UB) & 15, bt = idx / (16 * UB);
        const int b = bt * 16 + nn, j = j0 + mj;
        if (b >= B || j >= H) continue;
        float dh = d_out[((size_t)t * B + b) * H + j];
        const size_t zoff = ((size_t)dir * B + b) * H + j;
        if (!first) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) a += red[w][bt][mj][nn];
            dh += a + dhz[zoff];
        }
        const size_t row = ((size_t)t * B + b) * 2 + dir;
        const size_t gbase = row * 3 * H + j;
        const float r = G[gbase], z = G[gbase + H], n = G[gbase + 2 * H];
        const float gn = ghn[row * H + j];
        const float hp = has_prev ? hprev[(size_t)b * H + j] : 0.f;
        const float dn_pre = dh * (1.f - z) * (1.f - n * n);
        const float dz_pre = dh * (hp - n) * z * (1.f - z);
        const float dr_pre = dn_pre * gn * r * (1.f - r);
        G[gbase] = dr_pre;
        G[gbase + H] = dz_pre;
        G[gbase + 2 * H] = dn_pre;
        ghn[row * H + j] = dn_pre * r;
        dhz[zoff] = dh * z;
    }
}

template <int JU, int NBT>
void launch_fwd(float* G, float* ghn, float* hout, const float* w_hh, int T, int B, int H, hipStream_t st) {
    dim3 grid(ds2_cdiv(H, JU), 2), block(NW * 64);
    for (int s = 0; s < T; ++s)
        hipLaunchKernelGGL((gru_fwd_step_kernel<JU, NBT>), grid, block, 0, st, G, ghn, hout, w_hh, T, B, H, s);
}

template <int UB, int NBT>
void launch_bwd(float* G, float* ghn, const float* hout, const float* d_out, const float* w_hh_t, float* dhz,
                int T, int B, int H, hipStream_t st) {
    dim3 grid(ds2_cdiv(H, UB), 2), block(NW * 64);
    for (int s = 0; s < T; ++s)
        hipLaunchKernelGGL((gru_bwd_step_kernel<UB, NBT>), grid, block, 0, st, G, ghn, hout, d_out, w_hh_t, dhz,
                           T, B, H, s);
}

}  // namespace

extern "C" int ds2_gru_bidir_fwd(float* G, float* ghn, float* hout, const float* w_hh, int T, int B, int H,
                                 void* stream) {
    DS2_CHECK_ARG(G && ghn && hout && w_hh);
    DS2_CHECK_ARG(T > 0 && B > 0 && H > 0 && H % 8 == 0);
    DS2_CHECK_ARG(B <= 64);
    hipStream_t st = (hipStream_t)stream;
    if (B <= 16) launch_fwd<5, 1>(G, ghn, hout, w_hh, T, B, H, st);
    else if (B <= 32) launch_fwd<5, 2>(G, ghn, hout, w_hh, T, B, H, st);
    else launch_fwd<5, 4>(G, ghn, hout, w_hh, T, B, H, st);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" int ds2_gru_bidir_bwd(float* G, float* ghn, const float* hout, const float* d_out,
                                 const float* w_hh_t, float* dh_ws, int T, int B, int H, void* stream) {
    DS2_CHECK_ARG(G && ghn && hout && d_out && w_hh_t && dh_ws);
    DS2_CHECK_ARG(T > 0 && B > 0 && H > 0 && H % 8 == 0);
    DS2_CHECK_ARG(B <= 64);
    hipStream_t st = (hipStream_t)stream;
    if (B <= 16) launch_bwd<8, 1>(G, ghn, hout, d_out, w_hh_t, dh_ws, T, B, H, st);
    else if (B <= 32) launch_bwd<8, 2>(G, ghn, hout, d_out, w_hh_t, dh_ws, T, B, H, st);
    else launch_bwd<16, 4>(G, ghn, hout, d_out, w_hh_t, dh_ws, T, B, H, st);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

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

// ==========================================================================================================
// Persistent recurrence: ONE launch per layer pass, recurrent weights resident in VGPRs for all T steps.
//
// The launch-per-step kernels above re-stream W_hh (15.4 MB for both directions at H = 800) from the
// Infinity Cache every step because nothing keeps a workgroup's slice in its XCD's L2 across launches
// (rocprof: TCC_MISS ~ 18 MB per launch), which bounds a step at ~8 us.  Here every workgroup keeps its
// slice as MFMA A-operand fragments in registers (8 waves x KBW k-blocks of 16), so a step only moves the
// hidden state: each workgroup publishes its JU x B new values with write-through (sc1) stores, one lane
// adds to a per-direction arrival counter after every storing wave has drained (s_waitcnt vmcnt(0)) and
// the workgroup has met at a barrier, and consumers poll that counter with relaxed agent-scope loads,
// meet at a barrier, then read the full h_{t-1} with sc1 buffer loads straight into MFMA B-operand
// registers (cdna_hip_programming.md Guideline 16, counter form; MI355X_MICROARCH.md "Valid forms" row 1:
// one workgroup per CU, every handed-off byte stored AND loaded sc1).  All workgroups must be co-resident:
// grid = 2 * ceil(H/8) <= 240 workgroups of 512 threads.  Every spin is bounded by a wall-clock timeout
// that raises a flag the host checks (no hang on a lost workgroup).
// ==========================================================================================================
namespace {

constexpr int NWP = 8;                    // waves per persistent workgroup (2 per SIMD)
constexpr int PJU = 8;                    // hidden units per workgroup
constexpr unsigned long long SPIN_TICKS = 50000000ull;  // 0.5 s of the 100 MHz real-time counter

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NSHARD = 8;  // arrival counters per direction (workgroup x -> shard x % 8), each on its own 128-B line:
                           // 100 arrivals on ONE word serialise at ~12 ns each (MI355X_MICROARCH.md "fanin")
struct SyncWs {            // lives in caller-provided device memory, zeroed before every launch
    unsigned int arrive[2][NSHARD][32];
    unsigned int error;    // set to 1 on a spin timeout
};

__device__ __forceinline__ f32x4 load_sc1_b128(__amdgpu_buffer_rsrc_t rsrc, int byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 16 /* sc1 */);
    return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ void store_sc1(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave 0 only (all 64 lanes call it): lane l < NSHARD polls shard l until it holds step * (slices in that shard)
// arrivals; returns false on timeout.
__device__ __forceinline__ bool wait_arrivals(unsigned int* shards, int step, int nslice, int lane,
                                              unsigned int* err) {
    const unsigned int target = (unsigned int)step * (unsigned int)((nslice - lane + NSHARD - 1) / NSHARD);
    const bool poller = lane < NSHARD && lane < nslice;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        bool ok = true;
        if (poller)
            ok = __hip_atomic_load(shards + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target;
        if (__all(ok)) return true;
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_TICKS) {
            if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
}

template <int NBT, int KBW>
__global__ __launch_bounds__(NWP * 64) void gru_fwd_persistent_kernel(float* __restrict__ G, float* __restrict__ ghn,
                                                                      float* __restrict__ hout,
                                                                      const float* __restrict__ w_hh,
                                                                      SyncWs* __restrict__ sync, int T, int B, int H) {
    __shared__ float red[NWP][2][NBT][16][17];
    __shared__ int abort_flag;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dir = blockIdx.y, nslice = gridDim.x;
    const int j0 = blockIdx.x * PJU;
    const int m = lane & 15, q = lane >> 4;
    const int nkb = H >> 4;
    if (tid == 0) abort_flag = 0;

    // ---- resident weights: tile 0 rows = [r units | z units], tile 1 rows = [n units | unused]
    f32x4 wreg[2][KBW];
    {
        const int mj = m & 7, hi = m >> 3;
        const bool unit_ok = (j0 + mj) < H;
        const float* row0 = w_hh + ((size_t)dir * 3 * H + (size_t)(hi ? H : 0) + j0 + mj) * H;      // r or z
        const float* row1 = w_hh + ((size_t)dir * 3 * H + (size_t)2 * H + j0 + mj) * H;              // n
#pragma unroll
        for (int i = 0; i < KBW; ++i) {
            const int kb = wave + NWP * i;
            const bool ok = kb < nkb && unit_ok;
            const int k = kb * 16 + q * 4;
            wreg[0][i] = ok ? *reinterpret_cast<const f32x4*>(row0 + k) : f32x4{0.f, 0.f, 0.f, 0.f};
            wreg[1][i] = (ok && hi == 0) ? *reinterpret_cast<const f32x4*>(row1 + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // ---- gate-thread role
    const int jj = tid & 7, nn = (tid >> 3) & 15, gbt = tid >> 7;
    const int gb = gbt * 16 + nn, gj = j0 + jj;
    const bool gate_ok = (gbt < NBT) && (gb < B) && (gj < H);
    float hp = 0.f;                                     // this thread's h_{t-1}, carried in a register
    unsigned int* shards = &sync->arrive[dir][0][0];
    unsigned int* ctr = shards + (blockIdx.x % NSHARD) * 32;
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        const int t = dir == 0 ? s : T - 1 - s;
        float gi_r = 0.f, gi_z = 0.f, gi_n = 0.f;
        size_t gbase = 0;
        if (gate_ok) {                                  // independent of h: issue before the wait
            gbase = (((size_t)t * B + gb) * 2 + dir) * 3 * H + gj;
            gi_r = G[gbase];
            gi_z = G[gbase + H];
            gi_n = G[gbase + 2 * H];
        }
        if (s > 0) {
            if (wave == 0 && !wait_arrivals(shards, s, nslice, lane, &sync->error) && lane == 0) abort_flag = 1;
            __syncthreads();
            if (abort_flag) return;
            const int tprev = dir == 0 ? t - 1 : t + 1;
            const float* hprev = hout + ((size_t)dir * T + tprev) * B * H;
            const __amdgpu_buffer_rsrc_t rsrc =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hprev), 0, B * H * 4, 0x00020000);
#pragma unroll 1
            for (int bt = 0; bt < NBT; ++bt) {
                const int b = bt * 16 + m;
                f32x4 bf[KBW];
#pragma unroll
                for (int i = 0; i < KBW; ++i) {
                    const int kb = wave + NWP * i;
                    const bool ok = (kb < nkb) && (b < B);
                    bf[i] = ok ? load_sc1_b128(rsrc, (b * H + kb * 16 + q * 4) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
                f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
                for (int i = 0; i < KBW; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[0][i][e], bf[i][e], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[1][i][e], bf[i][e], acc1, 0, 0, 0);
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    red[wave][0][bt][4 * q + r][m] = acc0[r];
                    red[wave][1][bt][4 * q + r][m] = acc1[r];
                }
            }
        }
        __syncthreads();
        if (gate_ok) {
            float gh_r = 0.f, gh_z = 0.f, gh_n = 0.f;
            if (s > 0) {
#pragma unroll
                for (int w = 0; w < NWP; ++w) {
                    gh_r += red[w][0][gbt][jj][nn];
                    gh_z += red[w][0][gbt][8 + jj][nn];
                    gh_n += red[w][1][gbt][jj][nn];
                }
            }
            const float r = 1.f / (1.f + expf(-(gi_r + gh_r)));
            const float z = 1.f / (1.f + expf(-(gi_z + gh_z)));
            const float n = tanhf(gi_n + r * gh_n);
            const float h = (1.f - z) * n + z * hp;
            hp = h;
            store_sc1(&hout[(((size_t)dir * T + t) * B + gb) * H + gj], h);   // handed to every other workgroup
            G[gbase] = r;
            G[gbase + H] = z;
            G[gbase + 2 * H] = n;
            ghn[(((size_t)t * B + gb) * 2 + dir) * H + gj] = gh_n;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its stores
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int NBT, int KBW>
__global__ __launch_bounds__(NWP * 64) void gru_bwd_persistent_kernel(float* __restrict__ G, float* __restrict__ ghn,
                                                                      const float* __restrict__ hout,
                                                                      const float* __restrict__ d_out,
                                                                      const float* __restrict__ w_hh_t,
                                                                      SyncWs* __restrict__ sync, int T, int B, int H) {
    __shared__ float red[NWP][NBT][16][17];
    __shared__ int abort_flag;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dir = blockIdx.y, nslice = gridDim.x;
    const int j0 = blockIdx.x * PJU;
    const int m = lane & 15, q = lane >> 4;
    const int K = 3 * H, nkb = K >> 4;
    if (tid == 0) abort_flag = 0;

    f32x4 wreg[KBW];                                   // rows m < 8: column (j0+m) of W_hh = row of w_hh_t
    {
        const bool row_ok = (m < PJU) && (j0 + m < H);
        const float* row = w_hh_t + ((size_t)dir * H + j0 + (m & 7)) * K;
#pragma unroll
        for (int i = 0; i < KBW; ++i) {
            const int kb = wave + NWP * i;
            wreg[i] = (row_ok && kb < nkb) ? *reinterpret_cast<const f32x4*>(row + kb * 16 + q * 4)
                                           : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const int jj = tid & 7, nn = (tid >> 3) & 15, gbt = tid >> 7;
    const int gb = gbt * 16 + nn, gj = j0 + jj;
    const bool gate_ok = (gbt < NBT) && (gb < B) && (gj < H);
    float dhz = 0.f;                                    // dh * z carried to the next (earlier) step
    unsigned int* shards = &sync->arrive[dir][0][0];
    unsigned int* ctr = shards + (blockIdx.x % NSHARD) * 32;
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        const int t = dir == 0 ? T - 1 - s : s;
        const int tnext = dir == 0 ? t + 1 : t - 1;
        const bool has_prev = dir == 0 ? (t > 0) : (t < T - 1);
        // saved activations of step t (written by the forward pass, an earlier launch): plain loads
        float dh = 0.f, r = 0.f, z = 0.f, n = 0.f, gn = 0.f, hpv = 0.f;
        size_t row = 0, gbase = 0;
        if (gate_ok) {
            row = ((size_t)t * B + gb) * 2 + dir;
            gbase = row * 3 * H + gj;
            dh = d_out[((size_t)t * B + gb) * H + gj];
            r = G[gbase];
            z = G[gbase + H];
            n = G[gbase + 2 * H];
            gn = ghn[row * H + gj];
            if (has_prev) hpv = hout[(((size_t)dir * T + (dir == 0 ? t - 1 : t + 1)) * B + gb) * H + gj];
        }
        if (s > 0) {
            if (wave == 0 && !wait_arrivals(shards, s, nslice, lane, &sync->error) && lane == 0) abort_flag = 1;
            __syncthreads();
            if (abort_flag) return;
            // dGH of step tnext: [dr_pre | dz_pre] from G, d(gh_n) from ghn -- all stored sc1 by their owners
            const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(
                G + (size_t)tnext * B * 6 * H, 0, B * 6 * H * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_n = __builtin_amdgcn_make_buffer_rsrc(
                ghn + (size_t)tnext * B * 2 * H, 0, B * 2 * H * 4, 0x00020000);
#pragma unroll 1
            for (int bt = 0; bt < NBT; ++bt) {
                const int b = bt * 16 + m;
                // with several batch tiles the fragments are fetched in chunks to stay inside 256 VGPRs
                constexpr int CH = (NBT > 1 && KBW > 10) ? 10 : KBW;
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i0 = 0; i0 < KBW; i0 += CH) {
                    f32x4 bf[CH];
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const int i = i0 + c;
                        const int kb = wave + NWP * i;
                        const int k = kb * 16 + q * 4;
                        const bool ok = (i < KBW) && (kb < nkb) && (b < B);
                        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (ok) {
                            if (k < 2 * H) v = load_sc1_b128(rs_g, ((b * 2 + dir) * 3 * H + k) * 4);
                            else v = load_sc1_b128(rs_n, ((b * 2 + dir) * H + (k - 2 * H)) * 4);
                        }
                        bf[c] = v;
                    }
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        if (i0 + c < KBW) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[i0 + c][e], bf[c][e], acc, 0, 0, 0);
                        }
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) red[wave][bt][4 * q + rr][m] = acc[rr];
            }
        }
        __syncthreads();
        if (gate_ok) {
            if (s > 0) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < NWP; ++w) a += red[w][gbt][jj][nn];
                dh += a + dhz;
            }
            const float dn_pre = dh * (1.f - z) * (1.f - n * n);
            const float dz_pre = dh * (hpv - n) * z * (1.f - z);
            const float dr_pre = dn_pre * gn * r * (1.f - r);
            dhz = dh * z;
            store_sc1(&G[gbase], dr_pre);
            store_sc1(&G[gbase + H], dz_pre);
            store_sc1(&G[gbase + 2 * H], dn_pre);
            store_sc1(&ghn[row * H + gj], dn_pre * r);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

inline int pick_kbw(int need, const int* opts, int nopts) {
    for (int i = 0; i < nopts; ++i)
        if (opts[i] >= need) return opts[i];
    return -1;
}

template <int NBT>
bool launch_fwd_persistent(float* G, float* ghn, float* hout, const float* w_hh, SyncWs* sync, int T, int B, int H,
                           hipStream_t st) {
    const int opts[] = {1, 2, 4, 7};
    const int kbw = pick_kbw(ds2_cdiv(H / 16, NWP), opts, 4);
    dim3 grid(ds2_cdiv(H, PJU), 2), block(NWP * 64);
#define DS2_FWD_CASE(K)                                                                                              \
    case K:                                                                                                          \
        hipLaunchKernelGGL((gru_fwd_persistent_kernel<NBT, K>), grid, block, 0, st, G, ghn, hout, w_hh, sync, T, B, H); \
        return true;
    switch (kbw) {
        DS2_FWD_CASE(1)
        DS2_FWD_CASE(2)
        DS2_FWD_CASE(4)
        DS2_FWD_CASE(7)
    }
#undef DS2_FWD_CASE
    return false;
}

template <int NBT>
bool launch_bwd_persistent(float* G, float* ghn, const float* hout, const float* d_out, const float* w_hh_t,
                           SyncWs* sync, int T, int B, int H, hipStream_t st) {
    const int opts[] = {1, 2, 4, 8, 19};
    const int kbw = pick_kbw(ds2_cdiv(3 * H / 16, NWP), opts, 5);
    dim3 grid(ds2_cdiv(H, PJU), 2), block(NWP * 64);
#define DS2_BWD_CASE(K)                                                                                          \
    case K:                                                                                                      \
        hipLaunchKernelGGL((gru_bwd_persistent_kernel<NBT, K>), grid, block, 0, st, G, ghn, hout, d_out, w_hh_t, \
                           sync, T, B, H);                                                                       \
        return true;
    switch (kbw) {
        DS2_BWD_CASE(1)
        DS2_BWD_CASE(2)
        DS2_BWD_CASE(4)
        DS2_BWD_CASE(8)
        DS2_BWD_CASE(19)
    }
#undef DS2_BWD_CASE
    return false;
}

inline bool persistent_ok(int B, int H) {
    return (H % 16 == 0) && (2 * ds2_cdiv(H, PJU) <= 240) && (B <= 64) && (ds2_cdiv(3 * H / 16, NWP) <= 19) &&
           (ds2_cdiv(H / 16, NWP) <= 7);
}

}  // namespace

extern "C" size_t ds2_gru_sync_ws_bytes(void) { return sizeof(SyncWs); }

extern "C" int ds2_gru_bidir_fwd_persistent(float* G, float* ghn, float* hout, const float* w_hh, void* sync_ws,
                                            int T, int B, int H, void* stream) {
    DS2_CHECK_ARG(G && ghn && hout && w_hh && sync_ws);
    DS2_CHECK_ARG(T > 0 && B > 0 && H > 0);
    if (!persistent_ok(B, H)) {
        ds2_set_error("ds2_gru_bidir_fwd_persistent: unsupported shape B=%d H=%d", B, H);
        return DS2_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    SyncWs* sync = (SyncWs*)sync_ws;
    DS2_HIP(hipMemsetAsync(sync, 0, sizeof(SyncWs), st));
    bool ok;
    if (B <= 16) ok = launch_fwd_persistent<1>(G, ghn, hout, w_hh, sync, T, B, H, st);
    else if (B <= 32) ok = launch_fwd_persistent<2>(G, ghn, hout, w_hh, sync, T, B, H, st);
    else ok = launch_fwd_persistent<4>(G, ghn, hout, w_hh, sync, T, B, H, st);
    DS2_CHECK_ARG(ok);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" int ds2_gru_bidir_bwd_persistent(float* G, float* ghn, const float* hout, const float* d_out,
                                            const float* w_hh_t, void* sync_ws, int T, int B, int H, void* stream) {
    DS2_CHECK_ARG(G && ghn && hout && d_out && w_hh_t && sync_ws);
    DS2_CHECK_ARG(T > 0 && B > 0 && H > 0);
    if (!persistent_ok(B, H)) {
        ds2_set_error("ds2_gru_bidir_bwd_persistent: unsupported shape B=%d H=%d", B, H);
        return DS2_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    SyncWs* sync = (SyncWs*)sync_ws;
    DS2_HIP(hipMemsetAsync(sync, 0, sizeof(SyncWs), st));
    bool ok;
    if (B <= 16) ok = launch_bwd_persistent<1>(G, ghn, hout, d_out, w_hh_t, sync, T, B, H, st);
    else if (B <= 32) ok = launch_bwd_persistent<2>(G, ghn, hout, d_out, w_hh_t, sync, T, B, H, st);
    else ok = launch_bwd_persistent<4>(G, ghn, hout, d_out, w_hh_t, sync, T, B, H, st);
    DS2_CHECK_ARG(ok);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" int ds2_gru_persistent_supported(int B, int H) { return persistent_ok(B, H) ? 1 : 0; }

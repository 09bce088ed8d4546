// Persistent BiGRU recurrence kernels for gfx950 (one launch per layer pass).
#define DS2_PERSIST_MAIN_TU 1
#include "ds2_common.h"
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
#ifdef DS2_TIMING
// per-phase clock of one workgroup's thread 0 (tools/gru_phase_timing.py builds with -DDS2_TIMING=1)
__device__ long long ds2_tbuf[32 * 8];
#define DS2_TICK(i)                                                                                      \
    do {                                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x == 5 && blockIdx.y == 0 && blockIdx.z == 0 && s >= 100 && s < 132) \
            ds2_tbuf[(s - 100) * 8 + (i)] = __builtin_amdgcn_s_memtime();                                \
    } while (0)
extern "C" int ds2_debug_read_timing(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ds2_tbuf), sizeof(long long) * 32 * 8) == hipSuccess ? 0 : -1;
}
// round 4: per-WAVE stamps of one workgroup ([step 100..131][wave 8][stamp 12], lane 0 of every wave), both 4x4x1 kernels
// (the launch that ran last wins): tools/gru_wave_timing.py
__device__ long long ds2_wbuf[32 * 8 * 12];
#define DS2_WTICK(i)                                                                                      \
    do {                                                                                                  \
        if (lane == 0 && blockIdx.x == 5 && blockIdx.y == 0 && blockIdx.z == 0 && s >= 100 && s < 132)  \
            ds2_wbuf[((s - 100) * 8 + wave) * 12 + (i)] = __builtin_amdgcn_s_memtime();                   \
    } while (0)
extern "C" int ds2_debug_read_wave_timing(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ds2_wbuf), sizeof(long long) * 32 * 8 * 12) == hipSuccess ? 0 : -1;
}
__device__ unsigned int ds2_retries;            // fragments re-loaded by the canary protocol (all kernels)
extern "C" int ds2_debug_read_retries(unsigned int* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ds2_retries), sizeof(unsigned int)) != hipSuccess) return -1;
    if (reset) {
        const unsigned int z = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(ds2_retries), &z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#define DS2_RETRY_FLUSH(n) do { if ((threadIdx.x & 63) == 0 && (n)) __hip_atomic_fetch_add(&ds2_retries, (unsigned int)(n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
#else
#define DS2_TICK(i) do {} while (0)
#define DS2_WTICK(i) do {} while (0)
#define DS2_RETRY_FLUSH(n) do {} while (0)
#endif
#include "gru_persist_common.h"

namespace {

// ----------------------------------------------------------------------------------------------------------
// Backward recurrence on v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products per instruction).
//
// A workgroup owns 8 hidden units = 8 rows of w_hh_t: half a 16x16x4 tile, so half of every MFMA above is padding
// and, once the hand-off loads were made contiguous, MFMA issue (2400-deep K) became the largest part of the
// backward step.  In the 16-block form block kk (lanes 4kk..4kk+3) takes a DIFFERENT k, A = 4 units (one row
// group), B = 4 batch columns: one instruction retires 16 k x 4 units x 4 batch columns with no padding beyond
// rounding B up to a multiple of 4, at 10 cycles per instruction per wave with two waves per SIMD in flight
// (tools/attic/mfma4x4_probe.hip) against 32 for a 16x16x4.  Every block holds a partial sum over its own k: two DPP
// row_shr adds fold them inside each 16-lane row (lanes 12-15 hold the row's sum) and the gate threads add the
// remaining 4 rows x 8 waves from LDS.  k order: lane (kk, li) of wave w owns k = 64 G + 4 kk + e, G = w + 8 gi:
// one dwordx4 feeds the four instructions e = 0..3, for A (weights, resident) and B (the exchange ring) alike.
// Ring layout for this form: [dir][slot][batch quad][G][kk 16][batch rows of the quad][4 k] -- lane l of a
// wave-load reads bytes 16 l .. 16 l + 15 of one contiguous KB; a partial last quad (B & 3 rows) is stored compactly
// and the lanes of its missing rows load nothing (B = 10: 96 KB per workgroup per step instead of 115).
// ----------------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ float dpp_row_shr_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + N, 0xF, 0xF, true);
    return v + __int_as_float(t);
}

template <int N>
__device__ __forceinline__ float dpp_row_shl(float v) {   // lane l of a 16-lane row <- lane l + N (0 past the row)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + N, 0xF, 0xF, true));
}

constexpr int CGC = 4;   // batch quads per chunk = 16 batch columns
constexpr int RED4_PITCH = NWP * 4 + 4;   // floats per reduced value in LDS: 32 partials + 4 (bank spread, keeps 16-B alignment)

// NRG = row groups (of 4 units) per workgroup.  NRG = 2: 8 units, the whole batch.  NRG = 4 / 6: 16 / 24 units and a
// HALF / THIRD of the batch (blockIdx.z = batch part): the parts are independent recurrences, the workgroup count stays
// ~200, every CU holds 2x / 3x the weights (150 / 230 KB of its 512 KB register file) but pulls only its part's d(gh)
// per step and waits for 50 / 34 producers instead of 100.  The launcher picks the form with a cost model fitted to
// measurements: a step costs ~0.40 us per (batch quad x 8 units) of MFMA + fold work and ~0.34 us per 4 batch rows
// of hand-off loads per workgroup.
template <int NGI, int NRG, int PROTO, int NP = NRG / 2>
__global__ __launch_bounds__(NWP * 64) void gru_bwd_persistent4_kernel(float* __restrict__ G, float* __restrict__ ghn,
                                                                       const float* __restrict__ hout,
                                                                       const float* __restrict__ d_out,
                                                                       const float* __restrict__ w_hh_t,
                                                                       SyncWs* __restrict__ sync, float* __restrict__ ring,
                                                                       int T, int B, int H, int dbg, int spec) {
    // [value = (rg, cg, r 4, j 4)][RED4_PITCH: partial = wave * 4 + lane row]: a gate thread's 32 partials are
    // contiguous (eight ds_read_b128 in flight; as 32 scalar reads the compiler chained read -> wait -> add, ~0.6 us
    // per step), and the pitch of 36 puts the 16 storing lanes of a fold (4 rows x 4 j) on 16 different banks
    extern __shared__ __attribute__((aligned(16))) float red4[];
    __shared__ int abort_flag;
    constexpr int NPART = NP;                           // batch parts (NRG / 2, or 3 with NRG = 7: the 174-workgroup form)
    constexpr int UNITS = 4 * NRG;                      // hidden units per workgroup
    constexpr int CGW = NRG == 2 ? CGC : (NRG == 4 ? 2 : 1);   // batch quads per chunk (register budget)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y, part = blockIdx.z, nslice = gridDim.x;
    const int j0 = blockIdx.x * UNITS;
    const int kk = lane >> 2, li = lane & 3;
    const int K = 3 * H;
    const int bper = (B + NPART - 1) / NPART;           // batch rows per part (the last part may have fewer)
    const int b0 = part * bper;
    const int nb = min(bper, B - b0);                   // this workgroup's batch rows: b0 .. b0 + nb - 1
    const int ncg = (nb + 3) >> 2;
    const int ng = (K + 63) >> 6;                       // 64-wide k groups
    const int slot_floats = ng * 64 * nb;               // a partial last quad keeps only its nb & 3 rows
    // PROTO 0: counted, drained hand-off; 2: speculative loads, no per-step counters (see CANARY_BITS)
    constexpr bool SPEC = PROTO != 0, CAN = SPEC;
    constexpr int NSLOT = SPEC ? 4 : 2;
    constexpr int CAHEAD = 2;                           // the canary goes this many slots ahead of the payload
    // the wave that signals and polls: with the signal-first protocol the LAST wave, which has few or no gate threads (96 of
    // them in waves 0-1 at B = 10), so its arrival add does not queue behind its own hand-off stores
    constexpr int SIGW = CAN ? NWP - 1 : 0;
    float* my_ring = ring + (size_t)(dir * NPART + part) * NSLOT * ((size_t)ng * 64 * bper);
    if (nb <= 0) {                                      // an empty batch part: nobody waits for it
        if (tid == 0) leave_kernel(sync);
        return;
    }
    if (tid == 0) abort_flag = 0;

    f32x4 wA[NRG][NGI];                                 // weights of unit 4 rg + li at k = 64 G + 4 kk .. +3
#pragma unroll
    for (int rg = 0; rg < NRG; ++rg) {
        const int unit = j0 + 4 * rg + li;
        const float* row = w_hh_t + ((size_t)dir * H + (unit < H ? unit : 0)) * K;
#pragma unroll
        for (int gi = 0; gi < NGI; ++gi) {
            const int k = 64 * (wave + NWP * gi) + 4 * kk;
            wA[rg][gi] = (unit < H && k < K) ? *reinterpret_cast<const f32x4*>(row + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const int jj = tid % UNITS, nn = tid / UNITS;       // gate role: unit jj, local batch row nn
    const int gb = b0 + nn, gj = j0 + jj;
    const bool gate_ok = (nn < nb) && (gj < H);
    float dhz = 0.f;
    unsigned int* shards = &sync->arrive[dir][part][0][0];
    unsigned int* ctr = shards + (blockIdx.x % NSHARD) * 32;
    // this gate thread's three positions inside a ring slot (k index of unit j in gate g is g*H + j)
    const int rows4g = min(4, nb - (nn & ~3)) * 4;      // floats per k quad of this thread's batch quad
    const int hoff = (nn >> 2) * ng * 256 + (nn & 3) * 4;
    const int ho0 = hoff + (gj >> 2) * rows4g + (gj & 3), ho1 = hoff + ((H + gj) >> 2) * rows4g + ((H + gj) & 3),
              ho2 = hoff + ((2 * H + gj) >> 2) * rows4g + ((2 * H + gj) & 3);
    if (CAN && gate_ok) {                               // slot 0 (and 1) may hold an earlier launch's payload
#pragma unroll
        for (int sl = 0; sl < CAHEAD; ++sl) {
            store_canary(my_ring + (size_t)sl * slot_floats + ho0);
            store_canary(my_ring + (size_t)sl * slot_floats + ho1);
            store_canary(my_ring + (size_t)sl * slot_floats + ho2);
        }
    }
    if (CAN) wait_vmcnt0();
    __syncthreads();
    if (SPEC) {
        // The speculative protocol has no per-step counters, so nothing in a step tells a consumer that a producer has even
        // started: ONE counted rendezvous per launch makes sure every workgroup of the group has put its canaries into the
        // first slots before anybody reads them.
        if (tid == SIGW * 64) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wave == SIGW && !wait_arrivals(shards, 1, nslice, lane, &sync->error) && lane == 0) abort_flag = 1;
        __syncthreads();
        if (abort_flag) return;
    }
    int scur = 0, sprev = NSLOT - 1;                    // slot written this step / read this step (= last step's)
    int spec_delay = spec & 0xFF, spec_clean = 0, nretry = 0;   // speculative protocol: see spec_timing()

    // DS2_GI_PREFETCH (speculative protocol; see the forward kernel): the saved activations of step s + 1 are loaded right
    // behind the hand-off stores of step s, not at the top of step s + 1 in front of its hand-off loads
#ifndef DS2_GI_PREFETCH
#define DS2_GI_PREFETCH 1
#endif
    constexpr bool GIPF = SPEC && DS2_GI_PREFETCH;
    // Staged consumption (validate_fragments), measured round 4 at B = 10 / 8, us per step, all-at-once -> staged -> staged with
    // the loads of successive fragments 2 sleeps apart: FORWARD 2.77 -> 2.73 -> 2.67-2.72 / 2.40 -> 2.30 -> 2.28 (and with the
    // k-balanced deal 2.64 -> 2.66 -> 2.61 / 2.30 -> 2.30 -> 2.25; 4 sleeps: 2.56 / 2.17): on; BACKWARD, one stage per k group:
    // 2.92 -> 2.97 -> 3.0 / 2.41 -> 2.63 -> 2.50 (five validations per step cost more than the earlier start of the MFMAs
    // brings) -- two stages are what pays there (DS2_STAGED_BWD).
#ifndef DS2_STAGED_FWD
#define DS2_STAGED_FWD 1
#endif
#ifndef DS2_STAGED_BWD
#define DS2_STAGED_BWD 2         // 0 all at once, 1 one stage per k group, 2 TWO stages (k groups 0-2, then 3-4) -- shipped: with the
#endif                           // second stage's loads 4-8 sleeps behind the first's, B = 10 / 8 / 12: 2.88 -> 2.86 / 2.39 -> 2.32 / 2.95 -> 2.92
#ifndef DS2_STAGE_GAP_BWD
#define DS2_STAGE_GAP_BWD 6      // (swept: 0: 3.13 / 2.55, 4: 2.86 / 2.32, 8: 2.87 / 2.31, 12: 2.92 / 2.40, 16: 3.04 / 2.48 at B = 10 / 8)
#endif
#ifndef DS2_STAGE_GAP
#define DS2_STAGE_GAP 4          // (swept round 4, forward, B = 10 / 8 / 12: 2: 2.61 / 2.21 / -, 4: 2.56 / 2.17 / 2.57, 6: 2.58 / 2.16 / 2.58,
#endif                           //  8: 2.60 / 2.15 / 2.62, 10: 2.66 / 2.16 / 2.65, 14: 2.67 / 2.22 / 2.66)
    constexpr bool STAGED = SPEC && DS2_STAGED_BWD;        // validate + multiply k group by k group (validate_fragments)
    float dh = 0.f, r = 0.f, z = 0.f, n = 0.f, gn = 0.f, hpv = 0.f;
    // Running element offsets of this gate thread's saved activations: every array is affine in the time step, and the step
    // moves by tstep = -1 (direction 0 walks t = T - 1 .. 0) or + 1, so an offset advances by a CONSTANT per step -- four 64-bit
    // adds instead of the ~30 vector instructions of 64-bit multiplies the index expressions compiled to, on the two gate
    // waves, between a step's hand-off stores and its end-of-step barrier.
    const long long tstep = dir == 0 ? -1 : 1;
    const long long dG = tstep * B * 6 * H, dD = tstep * B * H, dN = tstep * B * 2 * H;
    const int t_first = dir == 0 ? T - 1 : 0;
    size_t of_g = (((size_t)t_first * B + gb) * 2 + dir) * 3 * H + gj;            // G[of_g + g H]: gate g of step t
    size_t of_d = ((size_t)t_first * B + gb) * H + gj;                            // d_out
    size_t of_n = (((size_t)t_first * B + gb) * 2 + dir) * H + gj;                // ghn
    size_t of_h = (((size_t)dir * T + t_first + tstep) * B + gb) * H + gj;        // hout of the step BEFORE t in forward time order
    for (int s = 0; s < T; ++s) {
        DS2_TICK(0);
        const int t = dir == 0 ? T - 1 - s : s;
        float sv_r = 0.f, sv_z = 0.f, sv_n = 0.f, sv_g = 0.f;
        // next != 0: the step after this one (its offsets = this step's + the constant strides)
        auto early_loads = [&](int next) {
            dh = r = z = n = gn = hpv = 0.f;
            if (gate_ok) {
                // saved activations (written by the forward pass, an earlier launch): plain loads
                const int tt = t + (next ? (int)tstep : 0);
                const bool has_prev = dir == 0 ? (tt > 0) : (tt < T - 1);
                const size_t g0 = of_g + (next ? dG : 0);
                dh = d_out[of_d + (next ? dD : 0)];
                r = G[g0];
                z = G[g0 + H];
                n = G[g0 + 2 * H];
                gn = ghn[of_n + (next ? dN : 0)];
                if (has_prev) hpv = hout[of_h + (next ? dD : 0)];
            }
        };
        // signal-first: the polling wave polls FIRST, then waits for its own (by then old) stores, and only then issues
        // these loads -- a vmcnt wait behind freshly issued HBM loads would put their latency on the step's chain
        const bool poll_first = CAN && !SPEC && wave == SIGW && s > 0;
        DS2_WTICK(0);
        if (!poll_first && !(GIPF && s > 0)) early_loads(0);
        DS2_WTICK(1);
        if (s > 0) {
            if (!SPEC) {
                if (!DS2_DBG(dbg, 1) && wave == SIGW && !wait_arrivals(shards, s, nslice, lane, &sync->error) && lane == 0)
                    abort_flag = 1;
                if (poll_first) {
                    wait_vmcnt0();   // the polling wave's deferred drain
                    early_loads(0);
                }
                DS2_TICK(1);
                __syncthreads();
                DS2_TICK(2);
                if (abort_flag) return;
            }
            const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
                my_ring + (size_t)sprev * slot_floats, 0, slot_floats * 4, 0x00020000);
            auto chunk = [&](int c, auto nci_tag) {
                constexpr int NCI = decltype(nci_tag)::value;
                f32x4 bf[NCI][NGI];
                f32x4 acc[NRG][NCI];
                if (SPEC && c == 0)                                // before the step's FIRST hand-off loads only
                    for (int i = 0; i < spec_delay; ++i) __builtin_amdgcn_s_sleep(1);
                auto load_frag = [&](int ci, int gi) {
                    const int g = wave + NWP * gi;                 // wave-uniform
                    const int cg = c * CGW + ci;
                    const int rows = min(4, nb - 4 * cg);          // batch rows this quad really has
                    // lanes of rows past the batch (and k groups past K) read nothing: out-of-range offset -> 0
                    // (k past K = 3H in the last group is never written by anyone and the ring is not cleared between
                    // launches: it must not be read either)
                    bf[ci][gi] = LOAD_HANDOFF(rs_x, (64 * g + 4 * kk < K && li < rows)
                                                        ? (cg * ng * 256 + ((g * 16 + kk) * rows + li) * 4) * 4
                                                        : OOB_OFFSET);
                };
#pragma unroll
                for (int gi = 0; gi < NGI; ++gi) {                 // k-group major: the MFMAs below consume in this order
                    if (STAGED && DS2_STAGE_GAP_BWD > 0 && (DS2_STAGED_BWD == 2 ? gi == (3 * NGI + 4) / 5 : gi > 0))
                        __builtin_amdgcn_s_sleep(DS2_STAGE_GAP_BWD);
#pragma unroll
                    for (int ci = 0; ci < NCI; ++ci) load_frag(ci, gi);
                }
                // every load of the chunk goes out before the first MFMA: left alone the scheduler sinks loads in between
                // the MFMAs and keeps only 2-5 in flight, which throttles a phase bounded by bytes in flight
                __builtin_amdgcn_sched_barrier(0);
                DS2_WTICK(2);
                auto products = [&]() {
#pragma unroll
                    for (int rg = 0; rg < NRG; ++rg)
#pragma unroll
                        for (int ci = 0; ci < NCI; ++ci) acc[rg][ci] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int gi = 0; gi < NGI; ++gi)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
                                for (int rg = 0; rg < NRG; ++rg)
                                    acc[rg][ci] = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[rg][gi][e], bf[ci][gi][e], acc[rg][ci], 0, 0, 0);
                };
                // validate BEFORE the matrix work: a failed attempt then costs one more round trip for the stale fragments only.
                // (Running the MFMAs optimistically first and repeating them after a re-load measured 15 % SLOWER per step.)
                if constexpr (STAGED) {
                    // staged: k group gi is validated right before its own MFMAs (see validate_fragments)
#pragma unroll
                    for (int rg = 0; rg < NRG; ++rg)
#pragma unroll
                        for (int ci = 0; ci < NCI; ++ci) acc[rg][ci] = f32x4{0.f, 0.f, 0.f, 0.f};
                    bool racc = false;
                    // DS2_STAGED_BWD = 1: one stage per k group; 2: two stages (k groups 0 .. SPLIT - 1, then the rest)
                    constexpr int SPLIT = (3 * NGI + 4) / 5;
                    constexpr int NSTG = DS2_STAGED_BWD == 2 && NGI > 1 ? 2 : NGI;
                    static_for<0, NSTG>([&](auto st_tag) {
                        constexpr int st = decltype(st_tag)::value;
                        constexpr int G0 = NSTG == NGI ? st : (st == 0 ? 0 : SPLIT), G1 = NSTG == NGI ? st + 1 : (st == 0 ? SPLIT : NGI);
                        validate_fragments<NCI, NGI, 0, G0, G1>(bf, load_frag, c == 0, spec, spec_delay, spec_clean, sync,
                                                                abort_flag, nretry, &racc, st == NSTG - 1);
                        if (st == 0) DS2_WTICK(3);
#pragma unroll
                        for (int gi = G0; gi < G1; ++gi)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
#pragma unroll
                                for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
                                    for (int rg = 0; rg < NRG; ++rg)
                                        acc[rg][ci] = __builtin_amdgcn_mfma_f32_4x4x1f32(wA[rg][gi][e], bf[ci][gi][e], acc[rg][ci], 0, 0, 0);
                    });
                } else {
                if (SPEC) validate_fragments<NCI, NGI>(bf, load_frag, c == 0, spec, spec_delay, spec_clean, sync, abort_flag, nretry);
                DS2_WTICK(3);
                products();
                }
                DS2_WTICK(4);
                // fold the 16 per-block partials: two DPP adds leave each 16-lane row's sum in its lanes 12..15
                // (all the DPP adds first, as independent chains, then ONE predicated region with the stores: written value
                // by value the compiler emits add_dpp / s_nop / mov_dpp / saveexec / add / ds_write / restore exec per value)
                float* rec = red4 + (lane & 3) * RED4_PITCH + wave * 4 + (lane >> 4);
                float fv[NRG][NCI][4];
#pragma unroll
                for (int rg = 0; rg < NRG; ++rg)
#pragma unroll
                    for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) fv[rg][ci][rr] = dpp_row_shr_add<4>(acc[rg][ci][rr]);
#pragma unroll
                for (int rg = 0; rg < NRG; ++rg)
#pragma unroll
                    for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) fv[rg][ci][rr] = dpp_row_shr_add<8>(fv[rg][ci][rr]);
                __builtin_amdgcn_sched_barrier(0);
                if ((lane & 15) >= 12) {
#pragma unroll
                    for (int rg = 0; rg < NRG; ++rg)
#pragma unroll
                        for (int ci = 0; ci < NCI; ++ci) {
                            const int cg = c * CGW + ci;
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) rec[((rg * ncg + cg) * 4 + rr) * 4 * RED4_PITCH] = fv[rg][ci][rr];
                        }
                }
            };
            if (SPEC && ncg == 1 && !DS2_DBG(dbg, 2)) {
                // the speculative protocol's case -- ONE batch quad per workgroup -- without the chunk loop's bookkeeping
                chunk(0, std::integral_constant<int, 1>{});
            } else if (!DS2_DBG(dbg, 2)) {
                const int nfull = ncg / CGW, tail = ncg - nfull * CGW;
#pragma unroll 1
                for (int c = 0; c < nfull; ++c) chunk(c, std::integral_constant<int, CGW>{});
                if constexpr (CGW > 1) {
                    if (tail == 1) chunk(nfull, std::integral_constant<int, 1>{});
                }
                if constexpr (CGW > 2) {
                    if (tail == 2) chunk(nfull, std::integral_constant<int, 2>{});
                    else if (tail == 3) chunk(nfull, std::integral_constant<int, 3>{});
                }
            }
        }
        DS2_TICK(3);
        DS2_WTICK(5);
        __syncthreads();
        DS2_TICK(4);
        DS2_WTICK(6);
        // (a wave that gave up on a payload that never came has raised abort_flag before the barrier above: read behind this
        // step's hand-off stores, tested at the end of the step -- see the forward kernel)
        if (gate_ok) {
            if (s > 0) {
                const int rg = jj >> 2, rr = jj & 3, cg = nn >> 2, bj = nn & 3;
                const f32x4* src = reinterpret_cast<const f32x4*>(red4 + (((rg * ncg + cg) * 4 + rr) * 4 + bj) * RED4_PITCH);
                f32x4 p[NWP];
#pragma unroll
                for (int w = 0; w < NWP; ++w) p[w] = src[w];
                const f32x4 q = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
                dh += ((q[0] + q[1]) + (q[2] + q[3])) + dhz;
            }
            const float dn_pre = dh * (1.f - z) * (1.f - n * n);
            const float dz_pre = dh * (hpv - n) * z * (1.f - z);
            const float dr_pre = dn_pre * gn * r * (1.f - r);
            dhz = dh * z;
            if (!CAN) {   // hand-off copies into the ring (write-through)
                float* slot = my_ring + (size_t)scur * slot_floats;
                store_sc1(slot + ho0, dr_pre);
                store_sc1(slot + ho1, dz_pre);
                store_sc1(slot + ho2, dn_pre * r);
            }
            sv_r = dr_pre;
            sv_z = dz_pre;
            sv_n = dn_pre;
            sv_g = dn_pre * r;
        }
        if (CAN) {
            // Hand-off stores of the speculative protocol: four neighbouring gate threads (units 4q .. 4q+3 of one batch row:
            // adjacent lanes, 16 contiguous bytes of the ring) hand their values to the first of them, which issues ONE
            // 16-byte write-through store per gate and ONE 16-byte canary store into the slot two ahead: 6 store instructions
            // of a quarter of the lanes instead of 12 of all of them.  (UNITS and H are multiples of 4: quads never straddle.)
            float q[3][4];
            // (the canary pattern is filtered out of the three OWN values, before they are handed around: 3 tests, not 12)
            const float pay_r = not_canary(sv_r), pay_z = not_canary(sv_z), pay_g = not_canary(sv_g);
#define DS2_QUAD_BCAST(J)                                                                                              \
    q[0][J] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(pay_r), (J) * 0x55, 0xF, 0xF, true));        \
    q[1][J] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(pay_z), (J) * 0x55, 0xF, 0xF, true));        \
    q[2][J] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(pay_g), (J) * 0x55, 0xF, 0xF, true));
            DS2_QUAD_BCAST(0)
            DS2_QUAD_BCAST(1)
            DS2_QUAD_BCAST(2)
            DS2_QUAD_BCAST(3)
#undef DS2_QUAD_BCAST
            // speculative protocol: the stores of the PREVIOUS step (a step old) are complete before this step's payload goes
            // out -- a consumer that has seen this payload can rely on every canary this workgroup wrote before it
            if (SPEC) wait_vmcnt0();
            DS2_WTICK(7);
            if (gate_ok && (jj & 3) == 0) {
                const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(my_ring, 0, NSLOT * slot_floats * 4,
                                                                                      0x00020000);
                const int sbase = scur * slot_floats, nbase = ((scur + CAHEAD) % NSLOT) * slot_floats;
                const int hos[3] = {ho0, ho1, ho2};
                const u32x4 can4 = {CANARY_BITS, CANARY_BITS, CANARY_BITS, CANARY_BITS};
                // (speculative protocol, fault-injection builds: workgroup 0 'loses' its payload of step 2 -> its consumers
                // must time out on the canary, not hang)
                const bool lose = SPEC && DS2_DBG(dbg, 64) && s == 2 && blockIdx.x + blockIdx.y + blockIdx.z == 0;
#pragma unroll
                for (int g3 = 0; g3 < 3; ++g3) {
                    const f32x4 v = {q[g3][0], q[g3][1], q[g3][2], q[g3][3]};
                    if (!lose) store_sc1_b128(rs_w, (sbase + hos[g3]) * 4, __builtin_bit_cast(u32x4, v));
                    store_sc1_b128(rs_w, (nbase + hos[g3]) * 4, can4);
                }
            }
        }
        const int aborted = CAN ? abort_flag : 0;      // (issued here, consumed at the end of the step)
        if (GIPF && s + 1 < T) early_loads(1);                                 // the next step's, behind this step's payload
        DS2_TICK(5);
        if (!CAN && !DS2_DBG(dbg, 4)) wait_vmcnt0();
        if (SPEC && (spec & (1 << 17))) wait_vmcnt0();   // self-timed: see spec_timing()
        sprev = scur;
        scur = scur == NSLOT - 1 ? 0 : scur + 1;
        DS2_TICK(6);
        DS2_WTICK(8);
        __syncthreads();
        DS2_TICK(7);
        DS2_WTICK(9);
        // (dbg bit 6, tests only: workgroup 0 'loses' its arrival of step 2 -> every waiter must time out, not hang)
        if (!SPEC && tid == SIGW * 64 && !(DS2_DBG(dbg, 64) && s == 2 && blockIdx.x + blockIdx.y + blockIdx.z == 0))
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gate_ok) {   // d(gi), d(gh_n) for the GEMMs that follow this launch: plain stores, off the critical path
            G[of_g] = sv_r;
            G[of_g + H] = sv_z;
            G[of_g + 2 * H] = sv_n;
            ghn[of_n] = sv_g;
        }
        of_g += dG;
        of_d += dD;
        of_n += dN;
        of_h += dD;
        // the drain, AFTER the signal: this step's stores (payload, next slot's canaries) are complete before the next
        // step's arrival add -- and before its early loads are issued, so the wait never covers a fresh HBM load.  The
        // polling wave (0) starts polling at once instead and drains after its poll has matched (its stores are old by then).
        if (CAN && !SPEC && wave != SIGW) wait_vmcnt0();
        DS2_WTICK(10);
        if (aborted) return;
    }
    DS2_RETRY_FLUSH(nretry);
    if (tid == SIGW * 64) leave_kernel(sync);   // the thread whose arrival adds must have been performed first
}

// ----------------------------------------------------------------------------------------------------------
// Backward recurrence, the BROADCAST deal (round 5; B = 5 .. 12: two or three batch parts of one quad, speculative hand-off).
//
// gru_bwd_persistent4_kernel gives every one of an instruction's 16 blocks a different k of the SAME 4 units, so each of a
// wave's NRG accumulators holds 16 partial sums per output: 2 DPP adds per value fold them inside the lane rows -- 40 / 48 /
// 56 DPP adds and 20 / 24 / 28 stores of a quarter of the lanes per wave and step (NRG = 5 / 6 / 7), between the last MFMA
// and the pre-gate barrier: 0.35 us on the waves that finish last (tools/gru_wave_timing.py).  Here the operands swap
// roles, as in the forward kernels: A = d(gates) (4 batch rows at one k; broadcast inside a group of blocks: CBSZ / ABID),
// B = weights, a different ROW GROUP of 4 units in every block of a group:
//   set A, CBSZ = 2: the 4 blocks of lane row g = row groups 0 .. 3 at k sub-index g                   (4 k per instruction)
//   set B, CBSZ = 1 (NRG >= 6): block pairs = row groups 4, 5 at k sub-index (g, q >> 1)               (8 k per instruction)
//   set C, CBSZ = 0 (NRG odd): all 16 blocks = the last row group at k sub-index (g, q)               (16 k per instruction)
// -- the same 5 / 6 / 7 instructions per 16 k, the same resident weights (20 NRG registers), the same exchange ring (lane
// (block, batch row) loads 4 consecutive k) -- and set A's partial sums over k sub-indices sit in different lane ROWS, which
// the gate threads add anyway with the 8 waves (32 partials per output, as before): no fold for 16 of the units, one DPP
// add per value for set B, two for set C (8 / 12 / 20 DPP adds per wave and step), and every lane stores its own 4 values.
// The gate role is spread over 4 threads per (unit, batch row): each adds 8 of the 32 partials (two ds_read_b128, not
// eight), two rotate-adds inside the 16-lane row give every one of the four the sum, all four do the (cheap) gate math, and
// the hand-off and saved-activation stores are dealt out among them: a storing lane issues 2 16-byte write-through stores
// (one gate's payload for 4 units + its canary) instead of 6, every lane ONE saved-activation store instead of 4.
// ----------------------------------------------------------------------------------------------------------
constexpr int RED5_PITCH = NWP * 4 + 4;
template <int N>
__device__ __forceinline__ float dpp_row_ror_add(float v) {   // v + (the value N lanes up the 16-lane row, cyclically)
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xF, 0xF, true);
    return v + __int_as_float(t);
}
template <int NGI, int NRG, int NPART = 3>
__global__ __launch_bounds__(NWP * 64) void gru_bwd_persistent5_kernel(float* __restrict__ G, float* __restrict__ ghn,
                                                                       const float* __restrict__ hout,
                                                                       const float* __restrict__ d_out,
                                                                       const float* __restrict__ w_hh_t,
                                                                       SyncWs* __restrict__ sync, float* __restrict__ ring,
                                                                       int T, int B, int H, int dbg, int spec) {
    static_assert(NRG >= 4 && NRG <= 7 && NPART >= 2 && NPART <= 3, "broadcast deal: 16 .. 28 units per workgroup, 2 or 3 batch parts");
    constexpr int UNITS = 4 * NRG;
    constexpr bool HASB = NRG >= 6, HASC = (NRG & 1) != 0;
    constexpr int NSLOT = 4, CAHEAD = 2, SIGW = NWP - 1;
    // [value = (row group, unit 4, batch row 4)][RED5_PITCH: 32 partials = (wave, lane row), rotated by 4 ((unit >> 1) + 2 rg)
    // so that the 32 lanes of a store instruction hit 16 banks twice]
    __shared__ __attribute__((aligned(16))) float red5[NRG * 16 * RED5_PITCH];
    __shared__ int abort_flag;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y, part = blockIdx.z, nslice = gridDim.x;
    const int j0 = blockIdx.x * UNITS;
    const int blk = lane >> 2, li = lane & 3, g = lane >> 4, q = (lane >> 2) & 3;
    const int K = 3 * H;
    const int bper = (B + NPART - 1) / NPART;
    const int b0 = part * bper;
    const int nb = min(bper, B - b0);                   // <= 4 (the launcher)
    const int ng = (K + 63) >> 6;
    const int slot_floats = ng * 64 * nb;
    float* my_ring = ring + (size_t)(dir * NPART + part) * NSLOT * ((size_t)ng * 64 * bper);
    if (nb <= 0) {
        if (tid == 0) leave_kernel(sync);
        return;
    }
    if (tid == 0) abort_flag = 0;

    // resident weights (B operands), component e of a register quad = k + e
    f32x4 wA[NGI][4], wB[HASB ? NGI : 1][2], wC[HASC ? NGI : 1];
    {
        auto wrow = [&](int rg) {
            const int unit = j0 + 4 * rg + li;
            return unit < H ? w_hh_t + ((size_t)dir * H + unit) * K : nullptr;
        };
        const float* rowA = wrow(q);
        const float* rowB = wrow(4 + (q & 1));
        const float* rowC = wrow(NRG - 1);
        auto ld = [&](const float* row, int k) {
            return (row && k < K) ? *reinterpret_cast<const f32x4*>(row + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        };
#pragma unroll
        for (int gi = 0; gi < NGI; ++gi) {
            const int k0 = 64 * (wave + NWP * gi);
#pragma unroll
            for (int a = 0; a < 4; ++a) wA[gi][a] = ld(rowA, k0 + 16 * g + 4 * a);
            if constexpr (HASB) {
#pragma unroll
                for (int a = 0; a < 2; ++a) wB[gi][a] = ld(rowB, k0 + 8 * (blk >> 1) + 4 * a);
            }
            if constexpr (HASC) wC[gi] = ld(rowC, k0 + 4 * blk);
        }
    }
    // gate role: lane = (unit in quad u4, gpart gp) inside a 16-lane row; a row = one unit quad (= row group) of one batch row
    const int u4 = tid & 3, gp = (tid >> 2) & 3, uq = (tid >> 4) % NRG, nn = (tid >> 4) / NRG;
    const int gb = b0 + nn, gj = j0 + 4 * uq + u4;
    const bool gate_ok = (nn < nb) && (gj < H);
    float dhz = 0.f;
    unsigned int* shards = &sync->arrive[dir][part][0][0];
    unsigned int* ctr = shards + (blockIdx.x % NSHARD) * 32;
    // hand-off: lane u4 == 0 of gpart gp < 3 stores gate gp of the quad's four units (k index of unit j in gate gp is gp H + j)
    const bool storer = gate_ok && u4 == 0 && gp < 3;
    const int hok = (gp < 3 ? gp : 0) * H + gj;
    const int ho = (nn & 3) * 4 + (hok >> 2) * (nb * 4) + (hok & 3);          // floats inside a slot (one batch quad)
    const u32x4 can4 = {CANARY_BITS, CANARY_BITS, CANARY_BITS, CANARY_BITS};
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(my_ring, 0, NSLOT * slot_floats * 4, 0x00020000);
    if (storer) {                                       // slot 0 (and 1) may hold an earlier launch's payload
#pragma unroll
        for (int sl = 0; sl < CAHEAD; ++sl) store_sc1_b128(rs_w, (sl * slot_floats + ho) * 4, can4);
    }
    wait_vmcnt0();
    __syncthreads();
    // one counted rendezvous per launch (see gru_bwd_persistent4_kernel)
    if (tid == SIGW * 64) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == SIGW && !wait_arrivals(shards, 1, nslice, lane, &sync->error) && lane == 0) abort_flag = 1;
    __syncthreads();
    if (abort_flag) return;
    int scur = 0, sprev = NSLOT - 1;
    int spec_delay = spec & 0xFF, spec_clean = 0, nretry = 0;

    // this lane's byte offsets inside a slot (rows past the batch and k past K load nothing) and its partial-sum positions
    int loff[NGI];
#pragma unroll
    for (int gi = 0; gi < NGI; ++gi) {
        const int gg = wave + NWP * gi;
        loff[gi] = (64 * gg + 4 * blk < K && li < nb) ? (((gg * 16 + blk) * nb + li) * 4) * 4 : OOB_OFFSET;
    }
    auto red_at = [&](int rg, int unit, int part32) {   // value (rg, unit, batch 0), partial position part32 (rotated)
        return red5 + ((rg * 4 + unit) * 4) * RED5_PITCH + ((part32 + 4 * ((unit >> 1) + 2 * rg)) & 31);
    };
    float* const redA = red_at(q, li, wave * 4 + g);
    float* const redB = red_at(4 + (q & 1), li, wave * 4 + g);
    float* const redC = red_at(NRG - 1, li, wave * 4 + g);
    const float* const red_r = red5 + ((uq * 4 + u4) * 4 + nn) * RED5_PITCH + 8 * gp;

    float dh = 0.f, r = 0.f, z = 0.f, n = 0.f, gn = 0.f, hpv = 0.f;
    const long long tstep = dir == 0 ? -1 : 1;
    const long long dG = tstep * B * 6 * H, dD = tstep * B * H, dN = tstep * B * 2 * H;
    const int t_first = dir == 0 ? T - 1 : 0;
    size_t of_g = (((size_t)t_first * B + gb) * 2 + dir) * 3 * H + gj;            // G[of_g + g H]: gate g of step t
    size_t of_d = ((size_t)t_first * B + gb) * H + gj;                            // d_out
    size_t of_n = (((size_t)t_first * B + gb) * 2 + dir) * H + gj;                // ghn
    size_t of_h = (((size_t)dir * T + t_first + tstep) * B + gb) * H + gj;        // hout of the step BEFORE t in forward time order
    for (int s = 0; s < T; ++s) {
        const int t = dir == 0 ? T - 1 - s : s;
        // next != 0: the step after this one (its offsets = this step's + the constant strides)
        auto early_loads = [&](int next) {
            dh = r = z = n = gn = hpv = 0.f;
            if (gate_ok) {
                const int tt = t + (next ? (int)tstep : 0);
                const bool has_prev = dir == 0 ? (tt > 0) : (tt < T - 1);
                const size_t g0 = of_g + (next ? dG : 0);
                dh = d_out[of_d + (next ? dD : 0)];
                r = G[g0];
                z = G[g0 + H];
                n = G[g0 + 2 * H];
                gn = ghn[of_n + (next ? dN : 0)];
                if (has_prev) hpv = hout[of_h + (next ? dD : 0)];
            }
        };
        DS2_WTICK(0);
        if (s == 0) early_loads(0);
        DS2_WTICK(1);
        if (s > 0 && !DS2_DBG(dbg, 2)) {                  // (ablation bits 2 / 2048 / 4096: see gru_fwd_persistent5_kernel)
            const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
                my_ring + (size_t)sprev * slot_floats, 0, slot_floats * 4, 0x00020000);
            f32x4 bf[1][NGI];
            for (int i = 0; i < spec_delay; ++i) __builtin_amdgcn_s_sleep(1);
            auto load_frag = [&](int, int gi) { bf[0][gi] = LOAD_HANDOFF(rs_x, loff[gi]); };
            constexpr int SPLIT = (3 * NGI + 4) / 5;       // two stages: k groups 0 .. SPLIT - 1, then the rest (DS2_STAGED_BWD = 2)
#pragma unroll
            for (int gi = 0; gi < NGI; ++gi) {
                if (DS2_STAGE_GAP_BWD > 0 && gi == SPLIT && NGI > 1) __builtin_amdgcn_s_sleep(DS2_STAGE_GAP_BWD);
                load_frag(0, gi);
            }
            __builtin_amdgcn_sched_barrier(0);             // every load out before the first MFMA
            DS2_WTICK(2);
            f32x4 accA[4], accB[2], accC;
#pragma unroll
            for (int a = 0; a < 4; ++a) accA[a] = f32x4{0.f, 0.f, 0.f, 0.f};
            accB[0] = accB[1] = accC = f32x4{0.f, 0.f, 0.f, 0.f};
            bool racc = false;
            constexpr int NSTG = NGI > 1 ? 2 : 1;
            static_for<0, NSTG>([&](auto st_tag) {
                constexpr int st = decltype(st_tag)::value;
                constexpr int G0 = st == 0 ? 0 : SPLIT, G1 = (NSTG == 1 || st == 1) ? NGI : SPLIT;
                if (!DS2_DBG(dbg, 2048))
                    validate_fragments<1, NGI, 0, G0, G1>(bf, load_frag, true, spec, spec_delay, spec_clean, sync, abort_flag,
                                                          nretry, &racc, st == NSTG - 1);
                if (st == 0) DS2_WTICK(3);
                if (!DS2_DBG(dbg, 4096))
#pragma unroll
                for (int gi = G0; gi < G1; ++gi)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        accA[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(bf[0][gi][e], wA[gi][0][e], accA[0], 2, 0, 0);
                        accA[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(bf[0][gi][e], wA[gi][1][e], accA[1], 2, 1, 0);
                        if constexpr (HASB) accB[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(bf[0][gi][e], wB[gi][0][e], accB[0], 1, 0, 0);
                        accA[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(bf[0][gi][e], wA[gi][2][e], accA[2], 2, 2, 0);
                        accA[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(bf[0][gi][e], wA[gi][3][e], accA[3], 2, 3, 0);
                        if constexpr (HASB) accB[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(bf[0][gi][e], wB[gi][1][e], accB[1], 1, 1, 0);
                        if constexpr (HASC) accC = __builtin_amdgcn_mfma_f32_4x4x1f32(bf[0][gi][e], wC[gi][e], accC, 0, 0, 0);
                    }
            });
            DS2_WTICK(4);
            // D register i = batch row i; lane (g, q, li): set A -> unit li of row group q, k sub-index g
            const f32x4 sa = (accA[0] + accA[1]) + (accA[2] + accA[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) redA[i * RED5_PITCH] = sa[i];
            if constexpr (HASB) {                          // k sub-indices (g, q >> 1): one DPP add, lanes 8 .. 15 of a row hold the sum
                const f32x4 sb = accB[0] + accB[1];
                float fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fb[i] = dpp_row_shr_add<8>(sb[i]);
                if (q >= 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) redB[i * RED5_PITCH] = fb[i];
                }
            }
            if constexpr (HASC) {                          // k sub-indices (g, q): two DPP adds, lanes 12 .. 15 hold the sum
                float fc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fc[i] = dpp_row_shr_add<8>(dpp_row_shr_add<4>(accC[i]));
                if (q == 3) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) redC[i * RED5_PITCH] = fc[i];
                }
            }
        }
        DS2_WTICK(5);
        __syncthreads();
        DS2_WTICK(6);
        float sv_r = 0.f, sv_z = 0.f, sv_n = 0.f, sv_g = 0.f;
        {
            float part = 0.f;
            if (s > 0 && nn < 4) {                        // this gpart's 8 of the 32 partials
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(red_r), p1 = *reinterpret_cast<const f32x4*>(red_r + 4);
                const f32x4 ps = p0 + p1;
                part = (ps[0] + ps[1]) + (ps[2] + ps[3]);
            }
            // the four gparts of a unit sit 4 lanes apart.  ror:8 FIRST: lanes l and l + 8 then hold identical bits, so the ror:4
            // step gives all four gparts (p0 + p2) + (p1 + p3) to the bit -- each keeps its own dh recurrence and stores a
            // different gate's output, which must come from ONE trajectory
            part = dpp_row_ror_add<4>(dpp_row_ror_add<8>(part));
            if (gate_ok) {
                dh += part + dhz;
                const float dn_pre = dh * (1.f - z) * (1.f - n * n);
                const float dz_pre = dh * (hpv - n) * z * (1.f - z);
                const float dr_pre = dn_pre * gn * r * (1.f - r);
                dhz = dh * z;
                sv_r = dr_pre;
                sv_z = dz_pre;
                sv_n = dn_pre;
                sv_g = dn_pre * r;
            }
        }
        {
            // the quad's four units of ONE gate (gpart gp's) to its first lane: one 16-byte payload, one 16-byte canary
            const float mine_v = not_canary(gp == 0 ? sv_r : (gp == 1 ? sv_z : sv_g));
            f32x4 v;
            v[0] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mine_v), 0x00, 0xF, 0xF, true));
            v[1] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mine_v), 0x55, 0xF, 0xF, true));
            v[2] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mine_v), 0xAA, 0xF, 0xF, true));
            v[3] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mine_v), 0xFF, 0xF, 0xF, true));
            // the stores of the PREVIOUS step (a step old) are complete before this step's payload goes out (see CANARY_BITS)
            wait_vmcnt0();
            DS2_WTICK(7);
            if (storer) {
                // (fault-injection builds: workgroup 0 'loses' its payload of step 2 -> its consumers time out, not hang)
                const bool lose = DS2_DBG(dbg, 64) && s == 2 && blockIdx.x + blockIdx.y + blockIdx.z == 0;
                if (!lose) store_sc1_b128(rs_w, (scur * slot_floats + ho) * 4, __builtin_bit_cast(u32x4, v));
                store_sc1_b128(rs_w, (((scur + CAHEAD) & (NSLOT - 1)) * slot_floats + ho) * 4, can4);
            }
        }
        const int aborted = abort_flag;                 // (issued here, consumed at the end of the step)
        const size_t og = of_g, on = of_n;
        if (s + 1 < T && !DS2_DBG(dbg, 8192)) early_loads(1);   // the next step's, behind this step's payload (bit 8192: see the forward kernel)
        if (spec & (1 << 17)) wait_vmcnt0();            // self-timed: see spec_timing()
        sprev = scur;
        scur = (scur + 1) & (NSLOT - 1);
        DS2_WTICK(8);
        __syncthreads();
        DS2_WTICK(9);
        if (gate_ok) {   // d(gi), d(gh_n) for the GEMMs that follow this launch: one plain store per lane, off the critical path
            if (gp == 3) ghn[on] = sv_g;
            else G[og + (size_t)gp * H] = gp == 0 ? sv_r : (gp == 1 ? sv_z : sv_n);
        }
        of_g += dG;
        of_d += dD;
        of_n += dN;
        of_h += dD;
        DS2_WTICK(10);
        if (aborted) return;
    }
    DS2_RETRY_FLUSH(nretry);
    if (tid == SIGW * 64) leave_kernel(sync);
}

// ----------------------------------------------------------------------------------------------------------
// Backward recurrence, the d(h) HAND-OFF (round 6; B = 5 .. 12 at H = 32 RPW): the broadcast deal above with a third of the
// bytes on the latency chain.
//
// Every backward kernel so far hands off d(gh) -- 3H values per batch row and step (K = 3H: gate g of unit j is k = g H + j).
// But d(gh)_t[b, g, j] = dh_t[b, j] * c_g[t, b, j] with
//     c_r = (1 - z)(1 - n^2) gh_n r (1 - r),   c_z = (h_prev - n) z (1 - z),   c_n = (1 - z)(1 - n^2) r
// -- functions of the FORWARD pass's saved activations alone, known before the backward pass starts.  So the ring carries
// dh_t (H values per row: the forward kernel's payload size), a consumer lane loads 4 consecutive units' dh once
// (write-through stores, sc1 loads: the chain) and multiplies it by the three gates' coefficient fragments, which it
// loaded ONE STEP AHEAD with plain cached loads from `coef` (T, B, 2, 3H) (written by the forward pass or by
// gru_bwd_coef_kernel; off the chain, L2-served: every workgroup of a (direction, part) reads the same 9.6 KB per row).
// The MFMAs, the partial-sum exchange through LDS, the four-lane gate role and the d(gi) / d(gh_n) outputs are the broadcast
// deal's; only the k order changes: a wave owns RPW = H / 32 RUNS of 4 consecutive units (ring slot = [run][batch row][4]),
// and an A register's 16 blocks are 16 (run, gate) ITEMS:
//     registers 3 f + g, f < RPW / 16:   block b = (run 16 f + b, gate g)   -- one dh load feeds three registers
//     the R = RPW % 16 last runs:        3 R items, gate-major, in ceil(3 R / 16) registers (one dh load each; the last
//                                        register's items sit in its first ceil(L / 4) block COLUMNS, so that set A skips
//                                        the instructions of the empty ones)
// H = 800: 5 registers (3 + 2), 3 dh loads + 5 coefficient loads per lane and step, 19 + 10 + 5 instructions x 4.
// ----------------------------------------------------------------------------------------------------------
template <int RPW>
struct DhDeal {
    static constexpr int FT = RPW / 16, R = RPW % 16;
    static constexpr int NREM = (3 * R + 15) / 16;
    static constexpr int NREG = 3 * FT + NREM, NLOAD = FT + NREM;
    static constexpr int L = 3 * R - 16 * (NREM - 1);           // items of the last register (R > 0)
    static constexpr int NCOL = R > 0 ? (L + 3) / 4 : 4;        // block columns the last register uses
    // set A / set B instruction (column a / pair member a) of register i: does any block it reads hold an item?
    static constexpr bool col_used(int i, int a) { return (R == 0 || i < NREG - 1) ? true : a < NCOL; }
    static constexpr int load_of(int i) { return i < 3 * FT ? i / 3 : FT + (i - 3 * FT); }
};
// item of (register i, block b): run inside the wave and gate, or false
template <int RPW>
__device__ __forceinline__ bool dh_item(int i, int b, int& run, int& gate) {
    using D = DhDeal<RPW>;
    if (i < 3 * D::FT) {
        run = 16 * (i / 3) + b;
        gate = i % 3;
        return true;
    }
    const int m = i - 3 * D::FT;
    int p;
    if (m < D::NREM - 1) {
        p = 16 * m + b;
    } else {
        const int a = b & 3, g = b >> 2;
        if (a >= D::NCOL) return false;
        const int pl = g * D::NCOL + a;
        if (pl >= D::L) return false;
        p = 16 * (D::NREM - 1) + pl;
    }
    gate = p / D::R;
    run = 16 * D::FT + p % D::R;
    return true;
}

// coef (T, B, 2, 3H) from the forward pass's saved tensors (G = (r, z, n), ghn = W_hn h, hout): an elementwise pass.  The
// forward persistent kernels that take a `coef` pointer write the same values from their gate threads instead.
__global__ __launch_bounds__(256) void gru_bwd_coef_kernel(const float* __restrict__ G, const float* __restrict__ ghn,
                                                           const float* __restrict__ hout, float* __restrict__ coef,
                                                           int T, int B, int H) {
    const size_t n4 = (size_t)T * B * 2 * (H / 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % (H / 4)) * 4;
        const size_t row = i / (H / 4);                        // (t, b, dir)
        const int dir = (int)(row & 1);
        const size_t tb = row >> 1;
        const int b = (int)(tb % B), t = (int)(tb / B);
        const float* g = G + row * 3 * H + j;
        const f32x4 r = *reinterpret_cast<const f32x4*>(g), z = *reinterpret_cast<const f32x4*>(g + H),
                    n = *reinterpret_cast<const f32x4*>(g + 2 * H);
        const f32x4 gn = *reinterpret_cast<const f32x4*>(ghn + row * H + j);
        const int tp = dir == 0 ? t - 1 : t + 1;
        f32x4 hp = {0.f, 0.f, 0.f, 0.f};
        if (tp >= 0 && tp < T) hp = *reinterpret_cast<const f32x4*>(hout + (((size_t)dir * T + tp) * B + b) * H + j);
        f32x4 cr, cz, cn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float an = (1.f - z[e]) * (1.f - n[e] * n[e]);
            cr[e] = an * gn[e] * r[e] * (1.f - r[e]);
            cz[e] = (hp[e] - n[e]) * z[e] * (1.f - z[e]);
            cn[e] = an * r[e];
        }
        float* c = coef + row * 3 * H + j;
        *reinterpret_cast<f32x4*>(c) = cr;
        *reinterpret_cast<f32x4*>(c + H) = cz;
        *reinterpret_cast<f32x4*>(c + 2 * H) = cn;
    }
}

#ifndef DS2_STAGE_GAP_BWD6
#define DS2_STAGE_GAP_BWD6 4
#endif
// ABL: ablation bits (results WRONG) as a COMPILE-TIME parameter, instantiated only in the fault-injection / timing builds and
// chosen by the launcher from DS2_GRU_DBG there -- 2: no hand-off loads and no MFMAs (the skeleton: the coefficient loads and
// multiplies stay), 2048: loads issued but not validated, 4096: no MFMAs, 8192: no prefetch of the next step's saved
// activations.  (As run-time branches they cost the UNABLATED launch of those builds 0.9 us per step: a branch around the dh
// loads makes the fragments two-path values, a branch per MFMA block splits the stream.)
template <int RPW, int NRG, int NPART = 3, int ABL = 0>
__global__ __launch_bounds__(NWP * 64) void gru_bwd_persistent6_kernel(float* __restrict__ G, float* __restrict__ ghn,
                                                                       const float* __restrict__ hout,
                                                                       const float* __restrict__ d_out,
                                                                       const float* __restrict__ w_hh_t,
                                                                       const float* __restrict__ coef,
                                                                       SyncWs* __restrict__ sync, float* __restrict__ ring,
                                                                       int T, int B, int H, int dbg, int spec) {
    static_assert(NRG >= 4 && NRG <= 7 && NPART >= 2 && NPART <= 3, "broadcast deal: 16 .. 28 units per workgroup, 2 or 3 batch parts");
    using D = DhDeal<RPW>;
    constexpr int NREG = D::NREG, NLOAD = D::NLOAD;
    constexpr int UNITS = 4 * NRG;
    constexpr bool HASB = NRG >= 6, HASC = (NRG & 1) != 0;
    constexpr int NSLOT = 4, CAHEAD = 2, SIGW = NWP - 1;
    constexpr bool NOLOAD = (ABL & 2) != 0, NOVAL = (ABL & (2048 | 2)) != 0, NOMFMA = (ABL & (4096 | 2)) != 0, NOPF = (ABL & 8192) != 0;
    __shared__ __attribute__((aligned(16))) float red5[NRG * 16 * RED5_PITCH];
    __shared__ int abort_flag;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y, part = blockIdx.z, nslice = gridDim.x;
    const int j0 = blockIdx.x * UNITS;
    const int blk = lane >> 2, li = lane & 3, g = lane >> 4, q = (lane >> 2) & 3;
    const int K = 3 * H;
    const int bper = (B + NPART - 1) / NPART;
    const int b0 = part * bper;
    const int nb = min(bper, B - b0);                   // <= 4 (the launcher)
    const int slot_floats = H * nb;                     // [run H / 4][batch row nb][4 units]
    float* my_ring = ring + (size_t)(dir * NPART + part) * NSLOT * ((size_t)H * bper);
    if (nb <= 0) {
        if (tid == 0) leave_kernel(sync);
        return;
    }
    if (tid == 0) abort_flag = 0;

    // resident weights (B operands): component e of a register quad = unit 4 run + e of the item's gate
    f32x4 wA[NREG][4], wB[HASB ? NREG : 1][2], wC[HASC ? NREG : 1];
    {
        auto wrow = [&](int rg) {
            const int unit = j0 + 4 * rg + li;
            return unit < H ? w_hh_t + ((size_t)dir * H + unit) * K : nullptr;
        };
        const float* rowA = wrow(q);
        const float* rowB = wrow(4 + (q & 1));
        const float* rowC = wrow(NRG - 1);
        auto ld = [&](const float* row, int i, int b) {
            int run, gate;
            if (!row || !dh_item<RPW>(i, b, run, gate)) return f32x4{0.f, 0.f, 0.f, 0.f};
            return *reinterpret_cast<const f32x4*>(row + gate * H + 4 * (RPW * wave + run));
        };
#pragma unroll
        for (int i = 0; i < NREG; ++i) {
#pragma unroll
            for (int a = 0; a < 4; ++a) wA[i][a] = ld(rowA, i, 4 * g + a);
            if constexpr (HASB) {
#pragma unroll
                for (int a = 0; a < 2; ++a) wB[i][a] = ld(rowB, i, (blk & ~1) + a);
            }
            if constexpr (HASC) wC[i] = ld(rowC, i, blk);
        }
    }
    // gate role: as gru_bwd_persistent5_kernel
    const int u4 = tid & 3, gp = (tid >> 2) & 3, uq = (tid >> 4) % NRG, nn = (tid >> 4) / NRG;
    const int gb = b0 + nn, gj = j0 + 4 * uq + u4;
    const bool gate_ok = (nn < nb) && (gj < H);
    float dhz = 0.f;
    unsigned int* shards = &sync->arrive[dir][part][0][0];
    unsigned int* ctr = shards + (blockIdx.x % NSHARD) * 32;
    // hand-off: the gp == 0 lane of a quad's first unit stores dh of the quad's four units -- ONE 16-byte payload per (run,
    // batch row) and step (the d(gh) hand-off: three)
    const bool storer = gate_ok && u4 == 0 && gp == 0;
    const int ho = ((gj >> 2) * nb + (nn & 3)) * 4;                             // floats inside a slot
    const u32x4 can4 = {CANARY_BITS, CANARY_BITS, CANARY_BITS, CANARY_BITS};
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(my_ring, 0, NSLOT * slot_floats * 4, 0x00020000);
    if (storer) {                                       // slot 0 (and 1) may hold an earlier launch's payload
#pragma unroll
        for (int sl = 0; sl < CAHEAD; ++sl) store_sc1_b128(rs_w, (sl * slot_floats + ho) * 4, can4);
    }
    wait_vmcnt0();
    __syncthreads();
    if (tid == SIGW * 64) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == SIGW && !wait_arrivals(shards, 1, nslice, lane, &sync->error) && lane == 0) abort_flag = 1;
    __syncthreads();
    if (abort_flag) return;
    int scur = 0, sprev = NSLOT - 1;
    int spec_delay = spec & 0xFF, spec_clean = 0, nretry = 0;

    // this lane's byte offsets: dh loads inside a ring slot, coefficient loads inside one time step's (B, 2, 3H) block
    int loff[NLOAD], coff[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        int run, gate;
        const bool ok = dh_item<RPW>(i, blk, run, gate) && li < nb;
        const int rg = RPW * wave + run;                // run of the row: units 4 rg .. 4 rg + 3
        // (a lane without an item, or past the batch, stages an element that exists: its dh fragment loads zeros and its weights
        // are zeros, so what it multiplies them by only has to be finite)
        coff[i] = ok ? ((((b0 + li) * 2 + dir) * 3 + gate) * H + 4 * rg) * 4 : ((b0 * 2 + dir) * 3 * H) * 4;
        if (i % 3 == 0 || i >= 3 * D::FT) loff[D::load_of(i)] = ok ? ((rg * nb + li) * 4) * 4 : OOB_OFFSET;
    }
    auto red_at = [&](int rg, int unit, int part32) {
        return red5 + ((rg * 4 + unit) * 4) * RED5_PITCH + ((part32 + 4 * ((unit >> 1) + 2 * rg)) & 31);
    };
    float* const redA = red_at(q, li, wave * 4 + g);
    float* const redB = red_at(4 + (q & 1), li, wave * 4 + g);
    float* const redC = red_at(NRG - 1, li, wave * 4 + g);
    const float* const red_r = red5 + ((uq * 4 + u4) * 4 + nn) * RED5_PITCH + 8 * gp;

    float dh = 0.f, r = 0.f, z = 0.f, n = 0.f, gn = 0.f, hpv = 0.f;
    const long long tstep = dir == 0 ? -1 : 1;
    const long long dG = tstep * B * 6 * H, dD = tstep * B * H, dN = tstep * B * 2 * H;
    const int t_first = dir == 0 ? T - 1 : 0;
    size_t of_g = (((size_t)t_first * B + gb) * 2 + dir) * 3 * H + gj;
    size_t of_d = ((size_t)t_first * B + gb) * H + gj;
    size_t of_n = (((size_t)t_first * B + gb) * 2 + dir) * H + gj;
    size_t of_h = (((size_t)dir * T + t_first + tstep) * B + gb) * H + gj;
    const int cstep_bytes = B * 6 * H * 4;             // one time step of coef
    // Coefficient fragments: plain cached loads into registers at the TOP of the step that consumes them, in front of the
    // first-attempt sleep and the dh loads.  An L2 / Infinity-Cache hit (0.1 - 0.25 us idle) returns under the sleep and the dh
    // loads' own round trip; nothing is carried around the loop, the compiler counts the loads itself (they are older than the
    // dh loads, so the vmcnt it emits for a dh fragment covers them).  Two other placements were built and measured first
    // (profiles/r06_recurrence_experiments.md): a step ahead into loop-carried registers (the compiler copies the tuples at
    // the loop's back edge behind a vmcnt(0): 3.0 us per step) and a step ahead by LDS-DMA (2.64 against the d(gh) form's 2.53:
    // five DMA issues of ~100 cycles each and the read-back sit in the matrix phase).
    auto load_coef1 = [&](__amdgpu_buffer_rsrc_t rs_c, int i) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_c, coff[i], 0, 0));
    };
    auto coef_rsrc = [&](int t) {                     // one time step of coef
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(coef) + (size_t)t * B * 6 * H, 0, cstep_bytes, 0x00020000);
    };
    auto early_loads = [&](int t, int next) {
        dh = r = z = n = gn = hpv = 0.f;
        if (gate_ok) {
            const int tt = t + (next ? (int)tstep : 0);
            const bool has_prev = dir == 0 ? (tt > 0) : (tt < T - 1);
            const size_t g0 = of_g + (next ? dG : 0);
            dh = d_out[of_d + (next ? dD : 0)];
            r = G[g0];
            z = G[g0 + H];
            n = G[g0 + 2 * H];
            gn = ghn[of_n + (next ? dN : 0)];
            if (has_prev) hpv = hout[of_h + (next ? dD : 0)];
        }
    };
    f32x4 cf[NREG];
    {
        const __amdgpu_buffer_rsrc_t rs_c0 = coef_rsrc(t_first);
#pragma unroll
        for (int i = 0; i < NREG; ++i) cf[i] = load_coef1(rs_c0, i);
        early_loads(t_first, 0);
    }
    int dead = 0;            // wave-uniform: a hand-off of this workgroup has timed out (the launch's results are void: finish quickly)
    // one time step; FIRST: no matrix phase (nothing has been handed off yet).  Returns true when the launch was aborted.
    auto step = [&](auto first_tag, const int s) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const int t = dir == 0 ? T - 1 - s : s;
        DS2_WTICK(0);
        DS2_WTICK(1);
        if constexpr (!FIRST) {
            const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
                my_ring + (size_t)sprev * slot_floats, 0, slot_floats * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_c = coef_rsrc(t);       // this step's rows: they go with this step's dh
            f32x4 bf[1][NLOAD];
            auto load_frag = [&](int, int l) { bf[0][l] = LOAD_HANDOFF(rs_x, loff[l]); };
            constexpr int SPLIT = D::FT > 0 && D::NREM > 0 ? D::FT : NLOAD;    // two stages: the full triples, then the last runs
            if constexpr (!NOLOAD) {
                for (int i = 0; i < spec_delay; ++i) __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int l = 0; l < NLOAD; ++l) {
                    if (DS2_STAGE_GAP_BWD6 > 0 && l == SPLIT && l > 0) __builtin_amdgcn_s_sleep(DS2_STAGE_GAP_BWD6);
                    load_frag(0, l);
                }
            } else {
#pragma unroll
                for (int l = 0; l < NLOAD; ++l) bf[0][l] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            __builtin_amdgcn_sched_barrier(0);             // every load out before the first MFMA
            DS2_WTICK(2);
            f32x4 accA[4], accB[2], accC;
#pragma unroll
            for (int a = 0; a < 4; ++a) accA[a] = f32x4{0.f, 0.f, 0.f, 0.f};
            accB[0] = accB[1] = accC = f32x4{0.f, 0.f, 0.f, 0.f};
            bool racc = false;
            constexpr int NSTG = SPLIT < NLOAD ? 2 : 1;
            static_for<0, NSTG>([&](auto st_tag) {
                constexpr int st = decltype(st_tag)::value;
                constexpr int L0 = st == 0 ? 0 : SPLIT, L1 = (NSTG == 1 || st == 1) ? NLOAD : SPLIT;
                // (a stage's validation stays behind the previous stage's MFMAs: hoisted into them, its vmcnt wait stalls the
                // wave in front of instructions whose operands have long arrived)
                if constexpr (st > 0) __builtin_amdgcn_sched_barrier(0);
                if constexpr (!NOVAL)
                    if (!dead)
                        validate_fragments<1, NLOAD, 0, L0, L1>(bf, load_frag, true, spec, spec_delay, spec_clean, sync, abort_flag,
                                                            nretry, &racc, st == NSTG - 1);
                if (st == 0) DS2_WTICK(3);
                auto reg_blocks = [&](auto mfma_tag) {
                    constexpr bool MFMA = decltype(mfma_tag)::value;
                    static_for<0, NREG>([&](auto i_tag) {
                        constexpr int i = decltype(i_tag)::value;
                        if constexpr (D::load_of(i) >= L0 && D::load_of(i) < L1) {
                            const f32x4 av = bf[0][D::load_of(i)] * cf[i];        // d(gh) of the lane's item: dh x coefficient
                            // this register's coefficients for the NEXT matrix phase (they go with THIS step's dh), as soon as
                            // it is free: the load runs under the MFMAs and the gate phase, a whole step ahead of its use
                            // (the ablations without MFMAs keep step 0's coefficients: with nothing to run under, these loads'
                            // latency would be what the shortened step measures)
                            if constexpr (MFMA) cf[i] = load_coef1(rs_c, i);
                            if constexpr (MFMA) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    accA[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[e], wA[i][0][e], accA[0], 2, 0, 0);
                                    if constexpr (D::col_used(i, 1)) accA[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[e], wA[i][1][e], accA[1], 2, 1, 0);
                                    if constexpr (HASB) accB[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[e], wB[i][0][e], accB[0], 1, 0, 0);
                                    if constexpr (D::col_used(i, 2)) accA[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[e], wA[i][2][e], accA[2], 2, 2, 0);
                                    if constexpr (D::col_used(i, 3)) accA[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[e], wA[i][3][e], accA[3], 2, 3, 0);
                                    if constexpr (HASB) accB[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[e], wB[i][1][e], accB[1], 1, 1, 0);
                                    if constexpr (HASC) accC = __builtin_amdgcn_mfma_f32_4x4x1f32(av[e], wC[i][e], accC, 0, 0, 0);
                                }
                            } else {
                                accC[0] += av[0] + av[1] + av[2] + av[3];         // (keeps the products alive)
                            }
                            // (one coefficient load per register block, in this order: left alone the compiler issues all of a
                            // stage's loads in one burst behind its MFMAs -- eight waves x 1 KB each at once back up the CU's one
                            // texture-address path and the waves stall at issue: matrix phase 0.86 -> 1.15 us in the stamps)
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                };
                reg_blocks(std::integral_constant<bool, !NOMFMA>{});
            });
            DS2_WTICK(4);
            const f32x4 sa = (accA[0] + accA[1]) + (accA[2] + accA[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) redA[i * RED5_PITCH] = sa[i];
            if constexpr (HASB) {
                const f32x4 sb = accB[0] + accB[1];
                float fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fb[i] = dpp_row_shr_add<8>(sb[i]);
                if (q >= 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) redB[i * RED5_PITCH] = fb[i];
                }
            }
            if constexpr (HASC) {
                float fc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fc[i] = dpp_row_shr_add<8>(dpp_row_shr_add<4>(accC[i]));
                if (q == 3) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) redC[i * RED5_PITCH] = fc[i];
                }
            }
        }
        DS2_WTICK(5);
        __syncthreads();
        DS2_WTICK(6);
        float sv_r = 0.f, sv_z = 0.f, sv_n = 0.f, sv_g = 0.f;
        {
            float part = 0.f;
            if constexpr (!FIRST) {
                if (nn < 4) {                             // this gpart's 8 of the 32 partials
                    const f32x4 p0 = *reinterpret_cast<const f32x4*>(red_r), p1 = *reinterpret_cast<const f32x4*>(red_r + 4);
                    const f32x4 ps = p0 + p1;
                    part = (ps[0] + ps[1]) + (ps[2] + ps[3]);
                }
                part = dpp_row_ror_add<4>(dpp_row_ror_add<8>(part));  // all four gparts: the same bits (see the broadcast deal)
            }
            if (gate_ok) {
                dh += part + dhz;
                const float dn_pre = dh * (1.f - z) * (1.f - n * n);
                const float dz_pre = dh * (hpv - n) * z * (1.f - z);
                const float dr_pre = dn_pre * gn * r * (1.f - r);
                dhz = dh * z;
                sv_r = dr_pre;
                sv_z = dz_pre;
                sv_n = dn_pre;
                sv_g = dn_pre * r;
            }
        }
        {
            // dh of the quad's four units to its first lane: one 16-byte payload, one 16-byte canary
            const float mine_v = not_canary(dh);
            f32x4 v;
            v[0] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mine_v), 0x00, 0xF, 0xF, true));
            v[1] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mine_v), 0x55, 0xF, 0xF, true));
            v[2] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mine_v), 0xAA, 0xF, 0xF, true));
            v[3] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mine_v), 0xFF, 0xF, 0xF, true));
            // The stores of the PREVIOUS step (a step old) are complete before this step's payload goes out (see CANARY_BITS).
            // By now they always are -- a wave's vector-memory operations complete in issue order, this step's dh loads were
            // issued behind them and the matrix phase has consumed those -- but the protocol's argument should not rest on
            // that property: the explicit wait stays.  It also covers the coefficient loads issued during the matrix phase,
            // L2 hits that have long returned: measured 2.68-2.73 with it against 2.68-2.71 us per step without (174
            // workgroups), 2.41-2.43 against 2.39-2.42 (240).
            wait_vmcnt0();
            DS2_WTICK(7);
            if (storer) {
                const bool lose = DS2_DBG(dbg, 64) && s == 2 && blockIdx.x + blockIdx.y + blockIdx.z == 0;
                if (!lose) store_sc1_b128(rs_w, (scur * slot_floats + ho) * 4, __builtin_bit_cast(u32x4, v));
                store_sc1_b128(rs_w, (((scur + CAHEAD) & (NSLOT - 1)) * slot_floats + ho) * 4, can4);
            }
        }
        const int aborted = abort_flag;                 // (issued here, consumed at the end of the step)
        const size_t og = of_g, on = of_n;
        // behind this step's payload: the next step's saved activations
        if (!NOPF && s + 1 < T) early_loads(t, 1);
        if (spec & (1 << 17)) wait_vmcnt0();            // self-timed: see spec_timing()
        sprev = scur;
        scur = (scur + 1) & (NSLOT - 1);
        DS2_WTICK(8);
        __syncthreads();
        DS2_WTICK(9);
        {   // d(gi), d(gh_n) for the GEMMs that follow this launch, off the critical path: the quad's four units of this lane's
            // plane (gp 0 .. 2: d(gi) of gates r, z, n; gp 3: d(gh_n)) gathered to its first lane -- one 16-byte store of a
            // quarter of the lanes per wave and step instead of a 4-byte store of all of them
            const float x = gp == 0 ? sv_r : (gp == 1 ? sv_z : (gp == 2 ? sv_n : sv_g));
            f32x4 v;
            v[0] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x00, 0xF, 0xF, true));
            v[1] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x55, 0xF, 0xF, true));
            v[2] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xAA, 0xF, 0xF, true));
            v[3] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xFF, 0xF, 0xF, true));
            if (gate_ok && u4 == 0) *reinterpret_cast<f32x4*>(gp == 3 ? ghn + on : G + og + (size_t)gp * H) = v;
        }
        of_g += dG;
        of_d += dD;
        of_n += dN;
        of_h += dD;
        DS2_WTICK(10);
        dead |= aborted;
    };
    // (no exit inside the loop: with one, the compiler routes it through the latch and copies every loop-carried register
    // tuple there, behind a vmcnt(0).  A workgroup whose hand-off timed out stops validating and runs its remaining steps
    // through -- its results are void anyway, the sticky error word is set -- instead of leaving)
    step(std::true_type{}, 0);
    for (int s = 1; s < T; ++s) step(std::false_type{}, s);
    if (dead) return;
    DS2_RETRY_FLUSH(nretry);
    if (tid == SIGW * 64) leave_kernel(sync);
}

// ----------------------------------------------------------------------------------------------------------
// Forward recurrence on v_mfma_f32_4x4x1_16b_f32 with A-operand broadcast (CBSZ / ABID; tools/attic/mfma4x4_bcast_probe.hip:
// blocks are grouped 2^CBSZ at a time and every block of a group takes its A rows from the group's block ABID).
//
// The 16x16x4 kernel above is MFMA-issue bound (tiles r|z and n|pad, 10 of 16 batch columns: 3328 pipe cycles per
// SIMD per step).  Here A = h (4 batch rows of one batch quad at ONE k), broadcast inside a group, and B = weights,
// a different row group (4 units of one gate) in every block of the group:
//   set A, CBSZ = 2: the 4 blocks of lane row g = row groups r0, r1, z0, z1 at k sub-index g      (4 k per instruction)
//   set B, CBSZ = 1: block pairs (g, p) = row groups n0, n1 at k sub-index (g, p)                 (8 k per instruction)
// so nothing is padded except B up to a multiple of 4, and -- unlike the plain 16-block form, where every block holds
// a different k of the SAME output and 72 values per wave need a DPP fold -- the partial sums over k sub-indices sit
// in different lane ROWS, which the gate threads add anyway together with the 8 waves (32 partials per output, read
// as b128).  One dwordx4 of h per lane (k = 64 G + 16 g + 4 q + e for lane (g, q, i), batch row i) feeds the
// 16 + 8 instructions (ABID = q resp. q & 1, e = 0..3) of a 64-wide k group: 24 MFMAs per batch quad per k group
// instead of 2 x 16 padded 16x16x4 ones at 3.2x the cycles each.  The exchange ring has the backward kernel's layout
// ([batch quad][G][kk = 4 g + q][rows][4 k], lane-contiguous, partial last quad compact).
// ----------------------------------------------------------------------------------------------------------
constexpr int FWD4_PITCH = NWP * 4 + 4;   // 32 partials (wave, lane row) + 4: 16-B aligned, rows spread over banks

// ---- the k-BALANCED deal (round 4; template parameter KPL > 0) ------------------------------------------------------
// The deal above hands k out in groups of 64 (one dwordx4 per lane feeds 16 instructions): H = 800 is 12.5 groups on 8 waves
// x NGI = 2 slots, so the busiest SIMD (waves 0 and 4) executes 4 groups' worth of MFMAs where the average is 3.125 -- and
// the matrix phase of a step ends with the busiest SIMD.  Here every one of the 32 lane rows (8 waves x 4) owns exactly
// KPL = H / 32 consecutive k: lane row R = 4 wave + g owns k = KPL R .. KPL R + KPL - 1, an instruction still retires ONE k
// per lane row (4 k per instruction, A broadcast from the block ABID names), so a wave issues KPL x NA instructions: 100
// instead of 128 at H = 800 with four A sets, the same on every SIMD.  Lane (g, q, .) supplies the A operand (h) of the
// instructions n = CH q .. CH q + CH - 1 of its row (CH = ceil(KPL / 4) consecutive k: 7 at KPL = 25; the last block of a
// row has what is left: 4), as components of NLD = ceil(CH / 4) dwordx4 loads.  Exchange ring slot:
// [batch quad][wave 8][ld NLD][kk = 4 g + q][rows][4]: lane l of a wave-load reads bytes 16 l .. 16 l + 15 of one contiguous
// KB as before.  Components past a lane's share (m >= CH: the same for every lane, a compile-time mask; whole loads past
// the last block's share: an out-of-range offset, i.e. zeros) belong to nobody, are never written and never tested or
// multiplied.  Selected when H % 32 == 0, the last block's share is whole loads, and there is no B set (launcher).

template <int Q>
__device__ __forceinline__ float dpp_quad_add(float v) {   // Q = quad_perm selector: 0xB1 = xor 1, 0x4E = xor 2
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), Q, 0xF, 0xF, true);
    return v + __int_as_float(t);
}

// P = batch parts (blockIdx.z), as in the backward kernel: a workgroup owns 8 P units and 1/P of the batch.  Its 6 P row
// groups (gate-major: rgi = gate * 2 P + unit group) are dealt, four at a time, to "A sets" (CBSZ = 2) and a last pair to
// a "B set" (CBSZ = 1) when 6 P is not a multiple of 4:  P = 1: (r0 r1 z0 z1) + (n0 n1);  P = 2: (r0-3) (z0-3) (n0-3);
// P = 3: (r0-3) (r4 r5 z0 z1) (z2-5) (n0-3) + (n4 n5).
// UGX != 0: UGX unit groups of 4 per workgroup instead of 2 P (the batch parts stay P).  UGX = 5 with P = 3: 20 units x 40
// slices x 3 parts x 2 directions = 240 workgroups -- the forward pass has the chip to itself, so it can take 240 CUs where
// the backward pass leaves 52 to the weight-gradient GEMMs; 15 row groups = (r0-3) (r4 z0 z1 z2) (z3 z4 n0 n1) (n2 n3 n4 -):
// the last A set's fourth block multiplies zeros (16 MFMAs per k instead of 18: 2.99 -> 2.9 us per step at B = 10).
template <int NGI, int P, int NBT, int PROTO, int UGX = 0, int KPL = 0>
__global__ __launch_bounds__(NWP * 64) void gru_fwd_persistent4_kernel(float* __restrict__ G, float* __restrict__ ghn,
                                                                       float* __restrict__ hout,
                                                                       const float* __restrict__ w_hh,
                                                                       SyncWs* __restrict__ sync, float* __restrict__ ring,
                                                                       int T, int B, int H, int dbg, int spec) {
    // [local batch row][gate row = gate * UNITS + unit][FWD4_PITCH partials]
    extern __shared__ __attribute__((aligned(16))) float red4[];
    __shared__ int abort_flag;
    constexpr int UG = UGX ? UGX : 2 * P, UNITS = 4 * UG, NRGI = 3 * UG, ROWS = 3 * UNITS;
    constexpr int NA = (NRGI % 4 == 3) ? NRGI / 4 + 1 : NRGI / 4, NBS = (NRGI % 4 == 2) ? 1 : 0;
    static_assert(NRGI % 4 != 1, "a single left-over row group is not dealt");
    constexpr int CGW = P == 1 ? CGC : (P == 2 ? 2 : 1);   // batch quads per chunk (register budget)
    constexpr int RPP = (NWP * 64) / (4 * UNITS);          // batch rows the gate role covers per pass (16, 8, 5)
    // k-balanced deal (see above): CH = k per (lane row, block), NLD = dwordx4 loads per lane, NFR = fragments per quad
    constexpr bool KB = KPL > 0;
    constexpr int CH = KB ? (KPL + 3) / 4 : 1, NLD = KB ? (CH + 3) / 4 : 1, NFR = KB ? NLD : NGI;
    static_assert(!KB || NBS == 0, "the k-balanced deal has no B set");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y, bpart = blockIdx.z, nslice = gridDim.x;
    const int j0 = blockIdx.x * UNITS;
    const int g = lane >> 4, q = (lane >> 2) & 3, li = lane & 3;     // lane row, block in row, row/col in block
    const int bper = (B + P - 1) / P;
    const int b0 = bpart * bper;
    const int nb = min(bper, B - b0);
    if (nb <= 0) {
        if (tid == 0) leave_kernel(sync);
        return;
    }
    const int ncg = (nb + 3) >> 2;
    const int ng = KB ? NWP * NLD : (H + 63) >> 6;            // 64-float groups per batch row of a ring slot
    const int slot_floats = ng * 64 * nb;
    constexpr bool SPEC = PROTO != 0, CAN = SPEC;             // protocols: see the backward kernel and CANARY_BITS
    constexpr int NSLOT = SPEC ? 4 : 2;
    constexpr int CAHEAD = 2;
    float* my_ring = ring + (size_t)(dir * P + bpart) * NSLOT * ((size_t)ng * 64 * bper);
    if (tid == 0) abort_flag = 0;

    // resident weights (B operands): set A a, block q = row group 4 a + q; set B: block q = pair q >> 1, row group 4 NA + (q & 1)
    f32x4 wA[KB ? 1 : NA][KB ? 1 : NGI][4], wB[NBS > 0 ? NBS : 1][KB ? 1 : NGI][2];
    float wK[NA][KB ? KPL : 1];                         // k-balanced deal: weight of row (4 a + q, li) at k = KPL R + n
    auto weight_row = [&](int rgi) {
        const int gate = rgi / UG, unit = j0 + 4 * (rgi % UG) + li;
        return (rgi < NRGI && unit < H) ? w_hh + ((size_t)dir * 3 * H + (size_t)gate * H + unit) * H : nullptr;
    };
    if constexpr (KB) {
        // (the launcher selects this deal only when 32 KPL == H: every lane row's KPL columns exist; dwordx4 loads from a
        // 4-byte aligned address -- as 100 scalar loads per lane, 64 different cache lines each, the preload took ~20 us)
        const int kbase = KPL * (4 * wave + g);
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const float* row = weight_row(4 * a + q);
#pragma unroll
            for (int n4 = 0; n4 < KPL / 4; ++n4) {
                const f32x4u v = row ? *reinterpret_cast<const f32x4u*>(row + kbase + 4 * n4) : f32x4u{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c) wK[a][4 * n4 + c] = v[c];
            }
#pragma unroll
            for (int n = KPL / 4 * 4; n < KPL; ++n) wK[a][n] = row ? row[kbase + n] : 0.f;
        }
    }
#pragma unroll
    for (int gi = 0; gi < (KB ? 0 : NGI); ++gi) {
        const int k0 = 64 * (wave + NWP * gi) + 16 * g;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const float* row = weight_row(4 * a + q);
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) {
                const int k = k0 + 4 * ab;
                wA[a][gi][ab] = (row && k < H) ? *reinterpret_cast<const f32x4*>(row + k) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if constexpr (NBS > 0) {
            const float* row = weight_row(4 * NA + (q & 1));
#pragma unroll
            for (int ab = 0; ab < 2; ++ab) {
                const int k = k0 + 8 * (q >> 1) + 4 * ab;
                wB[0][gi][ab] = (row && k < H) ? *reinterpret_cast<const f32x4*>(row + k) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    // gate role: 4 threads (gpart) per (local batch row nn [+ RPP bt], unit jj), each adds 8 of the 32 partials
    const int gpart = tid & 3, jj = (tid >> 2) % UNITS, nn = (tid >> 2) / UNITS;
    const int gj = j0 + jj;
    float hp[NBT];
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt) hp[bt] = 0.f;
    unsigned int* shards = &sync->arrive[dir][bpart][0][0];
    unsigned int* ctr = shards + (blockIdx.x % NSHARD) * 32;
    // this gate thread's position inside a ring slot, per batch pass (only gpart 0 stores)
    int hoff[NBT];
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt) {
        const int lb = bt * RPP + nn;
        const int rows4 = min(4, nb - (lb & ~3)) * 4;
        if constexpr (KB) {                             // unit gj = k: lane row R, instruction n, block qs, component m
            const int R = gj / KPL, n = gj % KPL, qs = n / CH, m = n % CH;
            hoff[bt] = (nn < RPP && lb < nb && gj < H && gpart == 0)
                           ? (lb >> 2) * ng * 256 + ((((R >> 2) * NLD + (m >> 2)) * 16 + 4 * (R & 3) + qs)) * rows4 + (lb & 3) * 4 + (m & 3)
                           : -1;
        } else
        hoff[bt] = (nn < RPP && lb < nb && gj < H && gpart == 0)
                       ? (lb >> 2) * ng * 256 + (gj >> 2) * rows4 + (lb & 3) * 4 + (gj & 3) : -1;
        if (CAN && hoff[bt] >= 0) {                     // slot 0 (and 1) may hold an earlier launch's payload
#pragma unroll
            for (int sl = 0; sl < CAHEAD; ++sl) store_canary(my_ring + (size_t)sl * slot_floats + hoff[bt]);
        }
    }
    if (CAN) wait_vmcnt0();
    __syncthreads();
    if (SPEC) {   // one counted rendezvous per launch: every producer's start-up canaries are in place before anybody reads
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wave == 0 && !wait_arrivals(shards, 1, nslice, lane, &sync->error) && lane == 0) abort_flag = 1;
        __syncthreads();
        if (abort_flag) return;
    }
    int scur = 0, sprev = NSLOT - 1;                    // slot written this step / read this step (= last step's)
    int spec_delay = spec & 0xFF, spec_clean = 0, nretry = 0;   // speculative protocol: see spec_timing()

    // DS2_GI_PREFETCH (speculative protocol): the gate pre-activations of step s + 1 are loaded right behind the hand-off stores
    // of step s instead of at the top of step s + 1 -- a wave's vector-memory operations complete in issue order, so loads
    // issued at a step's top stand in front of that step's hand-off loads, and they take slots of the CU's miss queue while
    // the hand-off loads are in flight (tools/gru_wave_timing.py: the second half of a step's hand-off loads lands a round
    // trip after the first)
#ifndef DS2_GI_PREFETCH
#define DS2_GI_PREFETCH 1
#endif
    constexpr bool GIPF = SPEC && DS2_GI_PREFETCH;
    constexpr bool STAGED = SPEC && DS2_STAGED_FWD;        // validate + multiply fragment by fragment (validate_fragments)
    float gi_r[NBT], gi_z[NBT], gi_n[NBT];
    for (int s = 0; s < T; ++s) {
        const int t = dir == 0 ? s : T - 1 - s;
        float sv_a[NBT], sv_g[NBT], sv_h[NBT];
        auto early_loads = [&](int t) {                 // independent of h: issued before the wait
#pragma unroll
            for (int bt = 0; bt < NBT; ++bt) {
                const int lb = bt * RPP + nn;
                gi_r[bt] = gi_z[bt] = gi_n[bt] = 0.f;
                if (nn < RPP && lb < nb && gj < H) {
                    const size_t gbase = (((size_t)t * B + b0 + lb) * 2 + dir) * 3 * H + gj;
                    gi_r[bt] = G[gbase];
                    gi_z[bt] = G[gbase + H];
                    gi_n[bt] = G[gbase + 2 * H];
                }
            }
        };
        // signal-first: the polling wave polls first, drains its (old) stores, then issues these loads (see the backward
        // kernel): a vmcnt wait must never sit behind freshly issued HBM loads
        const bool poll_first = CAN && !SPEC && wave == 0 && s > 0;
#pragma unroll
        for (int bt = 0; bt < NBT; ++bt) sv_a[bt] = sv_g[bt] = sv_h[bt] = 0.f;
        DS2_WTICK(0);
        if (!poll_first && !(GIPF && s > 0)) early_loads(t);
        DS2_WTICK(1);
        if (s > 0) {
            if (!SPEC) {
                if (!DS2_DBG(dbg, 1) && wave == 0 && !wait_arrivals(shards, s, nslice, lane, &sync->error) && lane == 0)
                    abort_flag = 1;
                if (poll_first) {
                    wait_vmcnt0();   // the polling wave's deferred drain
                    early_loads(t);
                }
                __syncthreads();
                if (abort_flag) return;
            }
            const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
                my_ring + (size_t)sprev * slot_floats, 0, slot_floats * 4, 0x00020000);
            auto chunk = [&](int c, auto nci_tag) {
                constexpr int NCI = decltype(nci_tag)::value;
                f32x4 bf[NCI][NFR];
                // accumulation chains per set and quad (by e parity): with four or more A sets one chain per set keeps the
                // matrix pipe fed (B = 10: 2.81 -> 2.77 us per step without the 16 adds that join two), below that two
                constexpr int NCH = NA >= 4 ? 1 : 2;
                f32x4 accA[NA][NCI][NCH], accB[NBS > 0 ? NBS : 1][NCI][NCH];
                if (SPEC && c == 0)                                // before the step's FIRST hand-off loads only
                    for (int i = 0; i < spec_delay; ++i) __builtin_amdgcn_s_sleep(1);
                // (timing builds: dbg bit 8 = waves 0-3 issue their hand-off loads ~0.2 us late, bit 9 = waves 4-7 -- which
                // half of the waves gets its data first?)
                if ((DS2_DBG(dbg, 256) && wave < 4) || (DS2_DBG(dbg, 512) && wave >= 4))
                    for (int i = 0; i < 8; ++i) __builtin_amdgcn_s_sleep(1);
                auto load_frag = [&](int ci, int gi) {
                    const int gq = wave + NWP * gi;                // wave-uniform k group
                    const int cg = c * CGW + ci;
                    const int rows = min(4, nb - 4 * cg);
                    const int kk = 4 * g + q;
                    if constexpr (KB) {
                        // gi = load index: components m = 4 gi .. 4 gi + 3 of this block's share (k = KPL R + CH q + m);
                        // a load wholly past the share (the row's last block) or past the batch rows reads nothing
                        bf[ci][gi] = LOAD_HANDOFF(rs_x, (CH * q + 4 * gi < KPL && 4 * gi < CH && li < rows)
                                                            ? (cg * ng * 256 + (((wave * NLD + gi) * 16 + kk) * rows + li) * 4) * 4
                                                            : OOB_OFFSET);
                        return;
                    }
                    // rows past the batch and k past H load nothing (out-of-range offset -> 0)
                    bf[ci][gi] = LOAD_HANDOFF(rs_x, (64 * gq + 4 * kk < H && li < rows)
                                                        ? (cg * ng * 256 + ((gq * 16 + kk) * rows + li) * 4) * 4
                                                        : OOB_OFFSET);
                };
#pragma unroll
                for (int gi = 0; gi < NFR; ++gi) {
                    if (STAGED && DS2_STAGE_GAP > 0 && gi > 0) __builtin_amdgcn_s_sleep(DS2_STAGE_GAP);
#pragma unroll
                    for (int ci = 0; ci < NCI; ++ci) load_frag(ci, gi);
                }
                __builtin_amdgcn_sched_barrier(0);                 // every load out before the first MFMA
                DS2_WTICK(2);
                auto products = [&]() {
                if constexpr (KB) {
#pragma unroll
                    for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
                        for (int h2 = 0; h2 < NCH; ++h2)
#pragma unroll
                            for (int a = 0; a < NA; ++a) accA[a][ci][h2] = f32x4{0.f, 0.f, 0.f, 0.f};
                    // instruction n: one k per lane row (k = KPL R + n), A = h from the row's block n / CH, component n % CH
                    static_for<0, KPL>([&](auto n_tag) {
                        constexpr int n = decltype(n_tag)::value, ABID = n / CH, m = n % CH;
#pragma unroll
                        for (int a = 0; a < NA; ++a)
#pragma unroll
                            for (int ci = 0; ci < NCI; ++ci)
                                accA[a][ci][n & (NCH - 1)] = __builtin_amdgcn_mfma_f32_4x4x1f32(
                                    bf[ci][m >> 2][m & 3], wK[a][n], accA[a][ci][n & (NCH - 1)], 2, ABID, 0);
                    });
                    return;
                }
                // two chains per set and quad (e parity): a wave needs ~12 independent chains to issue every 10 cycles
#pragma unroll
                for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
                    for (int h2 = 0; h2 < NCH; ++h2) {
#pragma unroll
                        for (int a = 0; a < NA; ++a) accA[a][ci][h2] = f32x4{0.f, 0.f, 0.f, 0.f};
                        accB[0][ci][h2] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                for (int gi = 0; gi < NGI; ++gi)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // ABID is an immediate; consecutive instructions go to different accumulators
#define DS2_FWD4_MFMA_A(ABID)                                                                                      \
    _Pragma("unroll") for (int a = 0; a < NA; ++a) _Pragma("unroll") for (int ci = 0; ci < NCI; ++ci)              \
        accA[a][ci][e & (NCH - 1)] = __builtin_amdgcn_mfma_f32_4x4x1f32(bf[ci][gi][e], wA[a][gi][ABID][e], accA[a][ci][e & (NCH - 1)], 2, ABID, 0);
#define DS2_FWD4_MFMA_B(ABID)                                                                                      \
    if constexpr (NBS > 0) {                                                                                       \
        _Pragma("unroll") for (int ci = 0; ci < NCI; ++ci) accB[0][ci][e & (NCH - 1)] =                            \
            __builtin_amdgcn_mfma_f32_4x4x1f32(bf[ci][gi][e], wB[0][gi][ABID][e], accB[0][ci][e & (NCH - 1)], 1, ABID, 0); \
    }
                        DS2_FWD4_MFMA_A(0)
                        DS2_FWD4_MFMA_B(0)
                        DS2_FWD4_MFMA_A(1)
                        DS2_FWD4_MFMA_B(1)
                        DS2_FWD4_MFMA_A(2)
                        DS2_FWD4_MFMA_A(3)
#undef DS2_FWD4_MFMA_A
#undef DS2_FWD4_MFMA_B
                    }
                };
                // validate BEFORE the matrix work: a failed attempt then costs one more round trip for the stale fragments only.
                // (Running the MFMAs optimistically first and repeating them after a re-load measured 15 % SLOWER per step.)
                if constexpr (STAGED) {
                    // staged: a fragment is validated right before its own MFMAs (see validate_fragments)
#pragma unroll
                    for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
                        for (int h2 = 0; h2 < NCH; ++h2) {
#pragma unroll
                            for (int a = 0; a < NA; ++a) accA[a][ci][h2] = f32x4{0.f, 0.f, 0.f, 0.f};
                            accB[0][ci][h2] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                    bool racc = false;
                    static_for<0, NFR>([&](auto fr_tag) {
                        constexpr int gi = decltype(fr_tag)::value;
                        validate_fragments<NCI, NFR, KB ? CH : 0, gi, gi + 1>(bf, load_frag, c == 0, spec, spec_delay, spec_clean, sync,
                                                                              abort_flag, nretry, &racc, gi == NFR - 1);
                        if (gi == 0) DS2_WTICK(3);
                        if constexpr (KB) {
                            static_for<0, KPL>([&](auto n_tag) {
                                constexpr int n = decltype(n_tag)::value, ABID = n / CH, m = n % CH;
                                if constexpr ((m >> 2) == gi) {
#pragma unroll
                                    for (int a = 0; a < NA; ++a)
#pragma unroll
                                        for (int ci = 0; ci < NCI; ++ci)
                                            accA[a][ci][n & (NCH - 1)] = __builtin_amdgcn_mfma_f32_4x4x1f32(
                                                bf[ci][m >> 2][m & 3], wK[a][n], accA[a][ci][n & (NCH - 1)], 2, ABID, 0);
                                }
                            });
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
#define DS2_FWD4_MFMA_A(ABID)                                                                                      \
    _Pragma("unroll") for (int a = 0; a < NA; ++a) _Pragma("unroll") for (int ci = 0; ci < NCI; ++ci)              \
        accA[a][ci][e & (NCH - 1)] = __builtin_amdgcn_mfma_f32_4x4x1f32(bf[ci][gi][e], wA[a][gi][ABID][e], accA[a][ci][e & (NCH - 1)], 2, ABID, 0);
#define DS2_FWD4_MFMA_B(ABID)                                                                                      \
    if constexpr (NBS > 0) {                                                                                       \
        _Pragma("unroll") for (int ci = 0; ci < NCI; ++ci) accB[0][ci][e & (NCH - 1)] =                            \
            __builtin_amdgcn_mfma_f32_4x4x1f32(bf[ci][gi][e], wB[0][gi][ABID][e], accB[0][ci][e & (NCH - 1)], 1, ABID, 0); \
    }
                                DS2_FWD4_MFMA_A(0)
                                DS2_FWD4_MFMA_B(0)
                                DS2_FWD4_MFMA_A(1)
                                DS2_FWD4_MFMA_B(1)
                                DS2_FWD4_MFMA_A(2)
                                DS2_FWD4_MFMA_A(3)
#undef DS2_FWD4_MFMA_A
#undef DS2_FWD4_MFMA_B
                            }
                        }
                    });
                } else {
                if constexpr (KB) {
                    if (SPEC) validate_fragments<NCI, NFR, CH>(bf, load_frag, c == 0, spec, spec_delay, spec_clean, sync, abort_flag, nretry);
                } else
                if (SPEC) validate_fragments<NCI, NGI>(bf, load_frag, c == 0, spec, spec_delay, spec_clean, sync, abort_flag, nretry);
                DS2_WTICK(3);
                products();
                }
                DS2_WTICK(4);
                // D register i = batch row i of the quad; lane (g, q, li): set A a -> gate row 4 (4 a + q) + li, k
                // sub-index g; set B -> gate row 4 (4 NA + (q & 1)) + li, k sub-indices (g, q >> 1): one DPP add folds
                // the two pairs
                const int partial = wave * 4 + g;
#pragma unroll
                for (int ci = 0; ci < NCI; ++ci) {
                    const int cg = c * CGW + ci;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float* rec = red4 + (size_t)((cg * 4 + i) * ROWS) * FWD4_PITCH + partial;
#pragma unroll
                        for (int a = 0; a < NA; ++a)
                            if (4 * (a + 1) <= NRGI || 4 * a + q < NRGI)       // (the padding block of a partial last set)
                                rec[(4 * (4 * a + q) + li) * FWD4_PITCH] = NCH == 2 ? accA[a][ci][0][i] + accA[a][ci][NCH - 1][i] : accA[a][ci][0][i];
                        if constexpr (NBS > 0) {
                            const float vb = dpp_row_shr_add<8>(NCH == 2 ? accB[0][ci][0][i] + accB[0][ci][NCH - 1][i] : accB[0][ci][0][i]);
                            if (q >= 2) rec[(4 * (4 * NA + (q & 1)) + li) * FWD4_PITCH] = vb;
                        }
                    }
                }
            };
            if (SPEC && ncg == 1 && !DS2_DBG(dbg, 2)) {
                // the speculative protocol's case -- ONE batch quad per workgroup -- without the chunk loop's bookkeeping
                chunk(0, std::integral_constant<int, 1>{});
            } else if (!DS2_DBG(dbg, 2)) {
                const int nfull = ncg / CGW, tail = ncg - nfull * CGW;
#pragma unroll 1
                for (int c = 0; c < nfull; ++c) chunk(c, std::integral_constant<int, CGW>{});
                if constexpr (CGW > 1) {
                    if (tail == 1) chunk(nfull, std::integral_constant<int, 1>{});
                }
                if constexpr (CGW > 2) {
                    if (tail == 2) chunk(nfull, std::integral_constant<int, 2>{});
                    else if (tail == 3) chunk(nfull, std::integral_constant<int, 3>{});
                }
            }
        }
        DS2_WTICK(5);
        __syncthreads();
        DS2_WTICK(6);
        // a wave gave up on a payload that never came (bounded re-loads): read with the partial sums, tested before the
        // stores (see the backward kernel)
        // (A wave that gave up on a payload that never came -- bounded re-loads -- has raised abort_flag before the barrier
        // above.  The flag is read behind this step's hand-off stores and tested at the END of the step: read up here, the
        // compiler made it wave-uniform and waited for its LDS round trip before it issued the partial-sum reads.  A step that
        // publishes a payload computed from a missing fragment harms nobody: the launch has failed, the sticky error flag is
        // set, the host raises.)
        // speculative protocol: the previous step's stores (a step old) are complete before this step's payload goes out
        if (SPEC) wait_vmcnt0();
        DS2_WTICK(7);
        // where this step's payload and the canary two slots ahead go (scalar arithmetic, off the path gate math -> store)
        float* const pay_slot = my_ring + (size_t)scur * slot_floats;
        float* const can_slot = my_ring + (size_t)((scur + CAHEAD) % NSLOT) * slot_floats;
#pragma unroll
        for (int bt = 0; bt < NBT; ++bt) {
            const int lb = bt * RPP + nn;
            const bool mine = nn < RPP && lb < nb && gj < H;
            float gh_r = 0.f, gh_z = 0.f, gh_n = 0.f;
            if (s > 0) {
                if (nn < RPP && lb < 4 * ncg) {          // this gpart's 8 partials (two b128) of the three gate rows
                    const f32x4* src = reinterpret_cast<const f32x4*>(red4 + (size_t)(lb * ROWS + jj) * FWD4_PITCH) + 2 * gpart;
                    constexpr int GSTR = UNITS * FWD4_PITCH / 4;     // f32x4 units between gates
                    const f32x4 r0 = src[0], r1 = src[1], z0 = src[GSTR], z1 = src[GSTR + 1], n0 = src[2 * GSTR],
                                n1 = src[2 * GSTR + 1];
                    const f32x4 rs = r0 + r1, zs = z0 + z1, ns = n0 + n1;
                    gh_r = (rs[0] + rs[1]) + (rs[2] + rs[3]);
                    gh_z = (zs[0] + zs[1]) + (zs[2] + zs[3]);
                    gh_n = (ns[0] + ns[1]) + (ns[2] + ns[3]);
                }
                gh_r = dpp_quad_add<0x4E>(dpp_quad_add<0xB1>(gh_r));
                gh_z = dpp_quad_add<0x4E>(dpp_quad_add<0xB1>(gh_z));
                gh_n = dpp_quad_add<0x4E>(dpp_quad_add<0xB1>(gh_n));
            }
            if (mine) {
                const float r = fast_sigmoid(gi_r[bt] + gh_r);
                const float z = fast_sigmoid(gi_z[bt] + gh_z);
                const float n = fast_tanh(gi_n[bt] + r * gh_n);
                const float h = (1.f - z) * n + z * hp[bt];
                hp[bt] = h;
                if (gpart == 0) {
                    // (speculative protocol, fault-injection builds: workgroup 0 'loses' its payload of step 2)
                    const bool lose = SPEC && DS2_DBG(dbg, 64) && s == 2 && blockIdx.x + blockIdx.y + blockIdx.z == 0;
                    if (!lose) store_sc1(pay_slot + hoff[bt], CAN ? not_canary(h) : h);
                    if (CAN) store_canary(can_slot + hoff[bt]);
                }
                sv_h[bt] = h;
                sv_a[bt] = gpart == 1 ? r : (gpart == 2 ? z : n);
                sv_g[bt] = gh_n;
            }
        }
        const int aborted = CAN ? abort_flag : 0;      // (issued here, consumed at the end of the step)
        if (GIPF && s + 1 < T) early_loads(dir == 0 ? s + 1 : T - 2 - s);             // the next step's, behind this step's payload
        if (!CAN && !DS2_DBG(dbg, 4)) wait_vmcnt0();   // every storing wave drains its hand-off store
        if (SPEC && (spec & (1 << 17))) wait_vmcnt0();   // self-timed: see spec_timing()
        sprev = scur;
        scur = scur == NSLOT - 1 ? 0 : scur + 1;
        DS2_WTICK(8);
        __syncthreads();
        DS2_WTICK(9);
        // (dbg bit 6, tests only: workgroup 0 'loses' its arrival of step 2 -> every waiter must time out, not hang)
        if (!SPEC && tid == 0 && !(DS2_DBG(dbg, 64) && s == 2 && blockIdx.x + blockIdx.y + blockIdx.z == 0))
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int bt = 0; bt < NBT; ++bt) {   // saved activations: read by later launches only, off the critical path
            const int lb = bt * RPP + nn;
            if (nn < RPP && lb < nb && gj < H) {
                const int gb = b0 + lb;
                const size_t gbase = (((size_t)t * B + gb) * 2 + dir) * 3 * H + gj;
                if (gpart == 0) {
                    hout[(((size_t)dir * T + t) * B + gb) * H + gj] = sv_h[bt];
                    ghn[(((size_t)t * B + gb) * 2 + dir) * H + gj] = sv_g[bt];
                } else {
                    G[gbase + (size_t)(gpart - 1) * H] = sv_a[bt];
                }
            }
        }
        // the drain, AFTER the signal (see CANARY_BITS): before the next step's arrival add and its early loads; the polling
        // wave drains after its poll instead
        if (CAN && !SPEC && wave != 0) wait_vmcnt0();
        DS2_WTICK(10);
        if (aborted) return;
    }
    DS2_RETRY_FLUSH(nretry);
    if (tid == 0) leave_kernel(sync);
}

// ----------------------------------------------------------------------------------------------------------
// Forward recurrence, the ROW deal (round 5): the 16 blocks of a v_mfma_f32_4x4x1_16b_f32 are 16 different ROW GROUPS of the
// workgroup's 3 UNITS <= 64 gate rows, ALL at the same k (A = h at ONE k, broadcast to the 16 blocks: CBSZ = 4, ABID picks the
// block of the A register that holds that k), so an instruction retires 1 k x 64 gate rows x 4 batch rows -- the same 256
// multiply-adds as in the deals above, KW = H / 8 instructions per wave (100 at H = 800: what the k-balanced deal issues) --
// and a wave's accumulator holds COMPLETE sums over the wave's KW columns: 8 partials per output (one per wave) instead of
// 32 (wave x lane row).  What that removes from the part of a step that has no hardware floor (last MFMA -> payload stores):
// 12 of a lane's 16 partial-sum stores to LDS (30.7 -> 7.7 KB per step through the 64-B/clk store path), three quarters of
// the gate threads' reads (six ds_read_b128 -> three ds_read_b64) and of their adds.
// Lane (b, li): B operand = w_hh row R = 4 b + li = lane (gate R / UNITS, unit R % UNITS) at k = KW wave + n for instruction n
// (KW registers, resident).  A operand: column n = 64 j + 4 b' + c of the wave sits in component c of dwordx4 load j, block
// b': lane (b', i) holds h[k = KW wave + 64 j + 4 b' + c][batch row i], instruction n reads register (j, c) with ABID = b'.
// Exchange ring slot: [wave 8][load NLD][block 16][rows][4] floats (the k-balanced deal's size; lane l of a wave-load reads
// bytes 16 l .. 16 l + 15 of one contiguous KB; blocks past the wave's KW columns are neither written nor loaded).  Four
// consecutive units of one batch row are 16 contiguous bytes: the first of their four gate threads stores them as ONE
// 16-byte write-through store (and ONE 16-byte canary two slots ahead) -- 20 + 20 store instructions of a sixteenth of the
// lanes per step where the deals above issue 80 + 80 dword stores (a dword sc1 store costs ~6x a dwordx4's time per byte:
// MI355X_MICROARCH.md, stores of each flavour), and every 64-byte segment of a slot is written by ONE workgroup.
// Speculative hand-off, one batch quad per workgroup, as gru_fwd_persistent4_kernel<.., PROTO = 2>.
// ----------------------------------------------------------------------------------------------------------
#ifndef DS2_FWD5_CHAINS
#define DS2_FWD5_CHAINS 4        // independent accumulation chains per wave (joined with NCH - 1 vector adds per batch row)
#endif
template <int KW, int UNITS, int P>
__global__ __launch_bounds__(NWP * 64) void gru_fwd_persistent5_kernel(float* __restrict__ G, float* __restrict__ ghn,
                                                                       float* __restrict__ hout,
                                                                       const float* __restrict__ w_hh,
                                                                       float* __restrict__ coef,
                                                                       SyncWs* __restrict__ sync, float* __restrict__ ring,
                                                                       int T, int B, int H, int dbg, int spec) {
    constexpr int ROWS = 3 * UNITS;                      // gate rows of the workgroup (<= 64)
    constexpr int NLD = (KW + 63) / 64;                  // dwordx4 loads per lane
    constexpr int NCH = DS2_FWD5_CHAINS;
    constexpr int NSLOT = 4, CAHEAD = 2;
    static_assert(ROWS <= 64 && KW % 4 == 0 && UNITS % 4 == 0 && NLD <= 2 && (NWP * 64) / (4 * UNITS) >= 4, "row deal: shape");
    // [batch row 4][gate row 64][8 partials, rotated by (row >> 2) so that the 32 lanes of a store hit 32 banks]
    __shared__ __attribute__((aligned(16))) float red5[4 * 64 * NWP];
    __shared__ int abort_flag;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y, bpart = blockIdx.z;
    const int nslice = gridDim.x;
    const int j0 = blockIdx.x * UNITS;
    const int blk = lane >> 2, li = lane & 3;
    const int bper = (B + P - 1) / P;
    const int b0 = bpart * bper;
    const int nb = min(bper, B - b0);                   // <= 4 (the launcher)
    if (nb <= 0) {
        if (tid == 0) leave_kernel(sync);
        return;
    }
    const int slot_floats = NWP * NLD * 64 * nb;
    float* my_ring = ring + (size_t)(dir * P + bpart) * NSLOT * ((size_t)NWP * NLD * 64 * bper);
    if (tid == 0) abort_flag = 0;

    // resident weights: row `lane` of the workgroup's gate rows, the wave's KW columns
    float wK[KW];
    {
        const int gate = lane / UNITS, unit = j0 + lane % UNITS;
        const float* row = (lane < ROWS && unit < H) ? w_hh + ((size_t)dir * 3 * H + (size_t)gate * H + unit) * H + KW * wave : nullptr;
#pragma unroll
        for (int n4 = 0; n4 < KW / 4; ++n4) {
            const f32x4u v = row ? *reinterpret_cast<const f32x4u*>(row + 4 * n4) : f32x4u{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) wK[4 * n4 + c] = v[c];
        }
    }
    // gate role: 4 threads (gpart) per (local batch row nn, unit jj), each adds 2 of the 8 partials of the three gate rows;
    // the 16 lanes of a row = 4 consecutive units (UNITS and the slices' first units are multiples of 4) of ONE batch row
    const int gpart = tid & 3, jj = (tid >> 2) % UNITS, nn = (tid >> 2) / UNITS;
    const int gj = j0 + jj;
    const bool mine = nn < nb && gj < H;
    const bool storer = mine && (tid & 15) == 0;        // hands off units gj .. gj + 3 of batch row nn (H % 4 == 0: all exist)
    float hp = 0.f;
    unsigned int* shards = &sync->arrive[dir][bpart][0][0];
    unsigned int* ctr = shards + (blockIdx.x % NSHARD) * 32;
    // where units gj .. gj + 3 (= columns k of every consumer) sit in a ring slot: wave kw, column n -> load n / 64, block n / 4 % 16
    int hoff = 0;
    const u32x4 can4 = {CANARY_BITS, CANARY_BITS, CANARY_BITS, CANARY_BITS};
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(my_ring, 0, NSLOT * slot_floats * 4, 0x00020000);
    if (storer) {
        const int kw = gj / KW, n = gj % KW;
        hoff = ((((kw * NLD + (n >> 6)) * 16 + ((n >> 2) & 15)) * nb + nn) * 4) * 4;      // bytes
#pragma unroll
        for (int sl = 0; sl < CAHEAD; ++sl) store_sc1_b128(rs_w, sl * slot_floats * 4 + hoff, can4);   // (an earlier launch's payload)
    }
    wait_vmcnt0();
    __syncthreads();
    // one counted rendezvous per launch: every producer's start-up canaries are in place before anybody reads
    if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 0 && !wait_arrivals(shards, 1, nslice, lane, &sync->error) && lane == 0) abort_flag = 1;
    __syncthreads();
    if (abort_flag) return;

    int scur = 0, sprev = NSLOT - 1;
    int spec_delay = spec & 0xFF, spec_clean = 0, nretry = 0;
    // this lane's byte offset inside a slot for load j (rows past the batch and blocks past the wave's columns load nothing)
    int loff[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j)
        loff[j] = (li < nb && 64 * j + 4 * blk < KW) ? ((((wave * NLD + j) * 16 + blk) * nb + li) * 4) * 4 : OOB_OFFSET;
    float* const red_w = red5 + lane * NWP + ((wave + (lane >> 2)) & (NWP - 1));
    // gate threads: partials 2 gpart, 2 gpart + 1 (rotated positions: any two, the four gparts cover all eight) of gate g
    const float* const red_r = red5 + (nn * 64 + jj) * NWP + 2 * gpart;
    // saved-activation / pre-activation element offsets (this thread's unit and batch row; advanced per step)
    const long long tstep = dir == 0 ? 1 : -1;
    const int t_first = dir == 0 ? 0 : T - 1;
    const long long dG = tstep * (long long)B * 6 * H, dO = tstep * (long long)B * H, dN = tstep * (long long)B * 2 * H;
    size_t of_g = (((size_t)t_first * B + b0 + nn) * 2 + dir) * 3 * H + gj;
    size_t of_o = (((size_t)dir * T + t_first) * B + b0 + nn) * H + gj;
    size_t of_n = (((size_t)t_first * B + b0 + nn) * 2 + dir) * H + gj;
    float gi_r = 0.f, gi_z = 0.f, gi_n = 0.f;
    if (mine) {
        gi_r = G[of_g];
        gi_z = G[of_g + H];
        gi_n = G[of_g + 2 * H];
    }
    for (int s = 0; s < T; ++s) {
        DS2_WTICK(0);
        DS2_WTICK(1);
        // where this step's payload and the canary two slots ahead go (scalar arithmetic, ahead of the matrix phase)
        const int pay_off = scur * slot_floats * 4 + hoff, can_off = ((scur + CAHEAD) & (NSLOT - 1)) * slot_floats * 4 + hoff;
        // (ablation bits, fault-injection / timing builds only -- results WRONG: 2 = no hand-off loads and no MFMAs (the step's
        // skeleton), 2048 = loads issued but not validated (nobody waits for anybody: matrix phase + skeleton), 4096 = no MFMAs
        // (hand-off + skeleton); bench.py's floor leg times them)
        if (s > 0 && !DS2_DBG(dbg, 2)) {
            const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
                my_ring + (size_t)sprev * slot_floats, 0, slot_floats * 4, 0x00020000);
            f32x4 bf[1][NLD];
            for (int i = 0; i < spec_delay; ++i) __builtin_amdgcn_s_sleep(1);
            auto load_frag = [&](int, int j) { bf[0][j] = LOAD_HANDOFF(rs_x, loff[j]); };
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                if (DS2_STAGE_GAP > 0 && j > 0) __builtin_amdgcn_s_sleep(DS2_STAGE_GAP);
                load_frag(0, j);
            }
            __builtin_amdgcn_sched_barrier(0);
            DS2_WTICK(2);
            f32x4 acc[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            bool racc = false;
            static_for<0, NLD>([&](auto j_tag) {
                constexpr int j = decltype(j_tag)::value;
                if (!DS2_DBG(dbg, 2048))
                    validate_fragments<1, NLD, 0, j, j + 1>(bf, load_frag, true, spec, spec_delay, spec_clean, sync, abort_flag,
                                                            nretry, &racc, j == NLD - 1);
                if (j == 0) DS2_WTICK(3);
                // component-major: consecutive instructions read different A registers and different accumulators
                if (!DS2_DBG(dbg, 4096))
                static_for<0, 64>([&](auto m_tag) {
                    constexpr int m = decltype(m_tag)::value, c = m >> 4, bb = m & 15, n = 64 * j + 4 * bb + c;
                    if constexpr (n < KW)
                        acc[m % NCH] = __builtin_amdgcn_mfma_f32_4x4x1f32(bf[0][j][c], wK[n], acc[m % NCH], 4, bb, 0);
                });
            });
            DS2_WTICK(4);
            f32x4 sum = acc[0];
            if constexpr (NCH == 2) sum = acc[0] + acc[1];
            if constexpr (NCH == 4) sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) red_w[i * 64 * NWP] = sum[i];
        }
        DS2_WTICK(5);
        __syncthreads();
        DS2_WTICK(6);
        // the previous step's stores (a step old) are complete before this step's payload goes out (see CANARY_BITS)
        // (it costs nothing measurable: 2.34 -> 2.34 us per step at B = 10 without it)
        wait_vmcnt0();
        DS2_WTICK(7);
        float gh_r = 0.f, gh_z = 0.f, gh_n = 0.f;
        if (s > 0) {
            if (nn < 4) {
                const f32x2 pr = *reinterpret_cast<const f32x2*>(red_r);
                const f32x2 pz = *reinterpret_cast<const f32x2*>(red_r + UNITS * NWP);
                const f32x2 pn = *reinterpret_cast<const f32x2*>(red_r + 2 * UNITS * NWP);
                gh_r = pr[0] + pr[1];
                gh_z = pz[0] + pz[1];
                gh_n = pn[0] + pn[1];
            }
            gh_r = dpp_quad_add<0x4E>(dpp_quad_add<0xB1>(gh_r));
            gh_z = dpp_quad_add<0x4E>(dpp_quad_add<0xB1>(gh_z));
            gh_n = dpp_quad_add<0x4E>(dpp_quad_add<0xB1>(gh_n));
        }
        float h = 0.f, sv_a = 0.f, sv_c = 0.f;
        float c_r = 0.f, c_z = 0.f, c_n = 0.f, c_hp = 0.f;
        if (mine) {
            const float r = fast_sigmoid(gi_r + gh_r);
            const float z = fast_sigmoid(gi_z + gh_z);
            const float n = fast_tanh(gi_n + r * gh_n);
            h = (1.f - z) * n + z * hp;
            sv_a = gpart == 1 ? r : (gpart == 2 ? z : n);
            c_r = r;
            c_z = z;
            c_n = n;
            c_hp = hp;
            hp = h;
        }
        {
            // the row's four units (lanes 0, 4, 8, 12 of the 16) to its first lane: one 16-byte payload, one 16-byte canary
            const float pay = not_canary(h);
            const f32x4 v = {pay, dpp_row_shl<4>(pay), dpp_row_shl<8>(pay), dpp_row_shl<12>(pay)};
            if (storer) {
                // (fault-injection builds: workgroup 0 'loses' its payload of step 2 -> its consumers time out, not hang)
                const bool lose = DS2_DBG(dbg, 64) && s == 2 && blockIdx.x + blockIdx.y + blockIdx.z == 0;
                if (!lose) store_sc1_b128(rs_w, pay_off, __builtin_bit_cast(u32x4, v));
                store_sc1_b128(rs_w, can_off, can4);
            }
        }
        const int aborted = abort_flag;                 // (issued here, consumed at the end of the step)
        // the next step's gate pre-activations, behind this step's payload (see DS2_GI_PREFETCH)
        const size_t og = of_g, oo = of_o, on = of_n;
        of_g += dG;
        of_o += dO;
        of_n += dN;
        // (ablation bit 8192: keep step 0's pre-activations -- with bits 2 + 8192 the step is so short that these loads' HBM
        // latency, which a full step never waits for, would be what the skeleton measurement shows)
        if (mine && s + 1 < T && !DS2_DBG(dbg, 8192)) {
            gi_r = G[of_g];
            gi_z = G[of_g + H];
            gi_n = G[of_g + 2 * H];
        }
        if (spec & (1 << 17)) wait_vmcnt0();            // self-timed: see spec_timing()
        sprev = scur;
        scur = (scur + 1) & (NSLOT - 1);
        DS2_WTICK(8);
        // (with the partial sums double-buffered this barrier is not needed for correctness -- and the step is 0.03-0.05 us
        // SLOWER without it, B = 10 / 12: 2.37 / 2.44 against 2.31-2.34 / 2.41: it keeps the eight waves' next hand-off loads together)
        __syncthreads();
        DS2_WTICK(9);
        {
            // Saved activations: read by later launches only, off the critical path.  Round 6: four consecutive units of one
            // plane and batch row = 16 contiguous bytes, gathered with DPP to the row's first unit's lane (as the payload is)
            // and stored as ONE dwordx4 -- gpart 0: hout (and gh_n), gparts 1 .. 3: the r / z / n planes (and the three
            // coefficient planes): two store instructions of a quarter of the lanes per wave and step, where dword stores
            // were four of three quarters of them (with the coefficient planes; they cost 2.12 -> 2.21 us per step that way).
            // The backward recurrence's coefficient planes (gru_bwd_persistent6_kernel; the expressions of gru_bwd_coef_kernel), behind
            // the end-of-step barrier with the saved-activation stores.  Their dozen instructions and one store are work in
            // front of the NEXT step's first-attempt sleep, which only has to shrink by as much -- but that sleep adapts by one
            // s_sleep per 2^k clean steps: with the launch's usual start (8 sleeps, k = 6) the planes cost 0.10 us per step
            // whichever half of them (the arithmetic, the store) was compiled in; started at 3 sleeps they cost 0.01
            // (launch_fwd_persistent5).
            if (coef) {
                __builtin_amdgcn_sched_barrier(0);
                if (mine) {
                    const float an = (1.f - c_z) * (1.f - c_n * c_n);
                    sv_c = gpart == 1 ? an * gh_n * c_r * (1.f - c_r) : (gpart == 2 ? (c_hp - c_n) * c_z * (1.f - c_z) : an * c_r);
                }
            }
            const float x1 = gpart == 0 ? h : sv_a, x2 = gpart == 0 ? gh_n : sv_c;
            const f32x4 v1 = {x1, dpp_row_shl<4>(x1), dpp_row_shl<8>(x1), dpp_row_shl<12>(x1)};
            const f32x4 v2 = {x2, dpp_row_shl<4>(x2), dpp_row_shl<8>(x2), dpp_row_shl<12>(x2)};
            if (mine && (tid & 12) == 0) {                // the first of the row's four units (all four exist: H % 4 == 0)
                float* const p1 = gpart == 0 ? hout + oo : G + og + (size_t)(gpart - 1) * H;
                *reinterpret_cast<f32x4*>(p1) = v1;
                if (gpart == 0) *reinterpret_cast<f32x4*>(ghn + on) = v2;
                else if (coef) *reinterpret_cast<f32x4*>(coef + og + (size_t)(gpart - 1) * H) = v2;
            }
        }
        DS2_WTICK(10);
        if (aborted) return;
    }
    DS2_RETRY_FLUSH(nretry);
    if (tid == 0) leave_kernel(sync);
}

// Speculative protocol: how long a wave waits (units of s_sleep 1 = 64 clocks) before a step's first hand-off loads, and
// between re-loads.  A failed attempt costs a full round trip and adds polling traffic from 1600 waves, so the first attempt
// is timed to land just after the slowest producer's payload; measured optimum (tools/gru_sweep.py, B = 4 .. 12): the forward
// kernel 8-12, the backward kernel (more and wider stores per producer) ~20.  DS2_GRU_SPEC_FWD / _BWD = "delay,backoff".
// def_inc / def_log2clean: the calling kernel's default adaptation policy (+inc sleeps after a step with a re-load, -1 after
// 2^log2clean clean ones); def_delay >= 0: its default first-attempt delay.  Round 5: the row-deal forward kernel wants a SLOW
// decay -- measured at B = 10 / 9 / 12, us per step, policy (1, 2) -> (1, 5) -> (1, 6): 2.33 / 2.20 / 2.40 -> 2.16 / 2.08 / 2.17
// -> 2.14 / 2.09 / 2.14 (a fixed delay of 6-8 sleeps without adaptation: 2.09-2.10) -- with the fast decay the delay creeps
// down until a wave re-loads (17 % of wave-steps did), and a re-load costs this kernel more than a few sleeps too many; the
// older kernels (B = 4 / 8) lose with the slow decay (forward 1.65 -> 1.75 / backward 2.32 -> 2.50) and keep (1, 2).
inline int spec_timing(int bwd, int def_delay = -1, int def_inc = 1, int def_log2clean = 2) {
    int delay = def_delay >= 0 ? def_delay : (bwd ? 14 : 10), backoff = 2;
    const char* e = ds2_tune_env(bwd ? "DS2_GRU_SPEC_BWD" : "DS2_GRU_SPEC_FWD");
    if (e) {
        delay = atoi(e);
        const char* c = strchr(e, ',');
        if (c) backoff = atoi(c + 1);
    }
    int adaptive = 1;                                  // bit 0: adapt the sleep count; bit 1: self-timed (wait for the own stores' acks)
    const char* a = ds2_tune_env("DS2_GRU_SPEC_ADAPT");
    if (a) adaptive = atoi(a);
    int inc = def_inc, log2clean = def_log2clean;      // adaptation: +inc after a step with a re-load, -1 after 2^log2clean clean ones
    const char* pol = ds2_tune_env("DS2_GRU_SPEC_POLICY");
    if (pol) {
        inc = atoi(pol);
        const char* c = strchr(pol, ',');
        if (c) log2clean = atoi(c + 1);
    }
    return (delay & 0xFF) | ((backoff & 0xFF) << 8) | ((adaptive & 3) << 16) | ((inc & 3) << 18) | ((log2clean & 7) << 20);
}

// The k-balanced deal of the forward 4x4x1 kernel (see the kernel): for which shapes it is built and selected.  KPL = H / 32
// k per lane row; the speculative single-quad case only (one batch quad per workgroup: B <= 12 with three parts, <= 8 with two);
// the row's last block must hold whole dwordx4 loads.  DS2_GRU_FWD_KBAL=0 switches it off (A/B timing).
inline int kbal_kpl(int H, int rows_per_part, int proto) {
    // Measured round 4 (B = 10 / 8, us per forward step, old deal -> balanced).  With round 3's step structure: 2.79 -> 2.81 /
    // 2.30 -> 2.28 -- the busiest SIMD issues 22 % fewer MFMAs and the step does not move, because the matrix phase of the waves
    // whose hand-off data lands first ran under the wait for the rest.  With the saved-activation loads prefetched, the
    // compiler's stray vmcnt(0) gone and staged consumption (all round 4) the matrix pipe is on the chain: 2.73 -> 2.66
    // (2.67-2.72 -> 2.61 with the 2-sleep gap between fragment loads) / 2.30 -> 2.30 (2.28 -> 2.25).
    const char* e = ds2_tune_env("DS2_GRU_FWD_KBAL");
    if (e && e[0] == '0') return 0;
    if (H % 32 != 0 || rows_per_part > 4 || proto == 0) return 0;
    const int kpl = H / 32, ch = (kpl + 3) / 4, last = kpl - 3 * ch;
    if (last < 0 || (last != ch && last % 4 != 0)) return 0;
    return kpl;
}

template <int P, int NBT, int PROTO, int UGX = 0>
bool launch_fwd_persistent4(float* G, float* ghn, float* hout, const float* w_hh, SyncWs* sync, float* ring, int T, int B,
                            int H, int dbg, hipStream_t st) {
    const int opts[] = {1, 2, 3};
    const int ngi = pick_kbw(ds2_cdiv(ds2_cdiv(H, 64), NWP), opts, 3);
    constexpr int UNITS = UGX ? 4 * UGX : 8 * P;
    dim3 grid(ds2_cdiv(H, UNITS), 2, P), block(NWP * 64);
    const int ncg = (ds2_cdiv(B, P) + 3) / 4;
    const size_t lds = (size_t)ncg * 4 * 3 * UNITS * FWD4_PITCH * sizeof(float);
    // k-balanced deal: built for H = 800 (KPL = 25), the BASELINE configs' width, in the forms without a B set
    if constexpr (PROTO != 0 && NBT == 1 && ((UGX ? 3 * UGX : 6 * P) % 4 != 2)) {
        if (kbal_kpl(H, ds2_cdiv(B, P), PROTO) == 25) {
            auto kern = &gru_fwd_persistent4_kernel<1, P, NBT, PROTO, UGX, 25>;
            if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return false;
            if (!grid_is_coresident(kern, grid, lds)) return false;
            // (this kernel's hand-off timing: DS2_GRU_FWD4_SPEC = "delay,inc,log2clean" for A/B runs)
            // (measured, us per forward step, (10, 1, 2) -> (8, 1, 5): B = 8 2.15 -> 2.05, B = 6 1.97 -> 1.94, B = 5 1.99 -> 1.94)
            int d = 8, inc = 1, l2c = 5;
            if (const char* e = ds2_tune_env("DS2_GRU_FWD4_SPEC")) sscanf(e, "%d,%d,%d", &d, &inc, &l2c);
            hipLaunchKernelGGL(kern, grid, block, lds, st, G, ghn, hout, w_hh, sync, ring, T, B, H, dbg, spec_timing(0, d, inc, l2c));
            return true;
        }
    }
#define DS2_FWD4_CASE(K)                                                                                         \
    case K:                                                                                                      \
        if (lds > 64 * 1024 &&                                                                                   \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_fwd_persistent4_kernel<K, P, NBT, PROTO, UGX>),           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)             \
            return false;                                                                                        \
        if (!grid_is_coresident(&gru_fwd_persistent4_kernel<K, P, NBT, PROTO, UGX>, grid, lds)) return false;                \
        hipLaunchKernelGGL((gru_fwd_persistent4_kernel<K, P, NBT, PROTO, UGX>), grid, block, lds, st, G, ghn, hout, w_hh, sync,    \
                           ring, T, B, H, dbg, spec_timing(0));                                                                  \
        return true;
    switch (ngi) {
        DS2_FWD4_CASE(1)
        DS2_FWD4_CASE(2)
        DS2_FWD4_CASE(3)
    }
#undef DS2_FWD4_CASE
    return false;
}

// the row deal (gru_fwd_persistent5_kernel): three batch parts of one quad each, 20 units per workgroup, H = 8 KW
template <int KW, int P = 3>
bool launch_fwd_persistent5(float* G, float* ghn, float* hout, const float* w_hh, float* coef, SyncWs* sync, float* ring, int T,
                            int B, int H, int dbg, hipStream_t st) {
    constexpr int UNITS = 20;
    dim3 grid(ds2_cdiv(H, UNITS), 2, P), block(NWP * 64);
    auto kern = &gru_fwd_persistent5_kernel<KW, UNITS, P>;
    if (!grid_is_coresident(kern, grid, 0)) return false;
    // (first-attempt delay and adaptation policy of THIS kernel: DS2_GRU_FWD5_SPEC = "delay,inc,log2clean" for A/B runs)
    // With the coefficient planes as an output (a training pass whose backward recurrence hands off dh) a step has more work
    // behind its payload store and wants a shorter sleep from the start -- measured at B = 10 / 9, us per forward step with the
    // planes, (8, 1, 6) -> (3, 1, 5): 2.27 / 2.16 -> 2.16 / 2.08 (without the planes: 2.15 at B = 10 either way); B = 12 (three full
    // quads) is the exception: 2.32 -> 2.48
    int d = 8, inc = 1, l2c = 6;
    if (B <= 11) d = 3, l2c = 5;                    // (without the planes: 2.18 -> 2.15 at B = 10)
    if (const char* e = ds2_tune_env("DS2_GRU_FWD5_SPEC")) sscanf(e, "%d,%d,%d", &d, &inc, &l2c);
    hipLaunchKernelGGL(kern, grid, block, 0, st, G, ghn, hout, w_hh, coef, sync, ring, T, B, H, dbg, spec_timing(0, d, inc, l2c));
    return true;
}

template <int NRG, int PROTO, int NP = NRG / 2>
bool launch_bwd_persistent4(float* G, float* ghn, const float* hout, const float* d_out, const float* w_hh_t,
                            SyncWs* sync, float* ring, int T, int B, int H, int dbg, hipStream_t st) {
    const int opts[] = {1, 2, 3, 5};
    const int ngi = pick_kbw(ds2_cdiv(ds2_cdiv(3 * H, 64), NWP), opts, 4);
    constexpr int NPART = NP;
    dim3 grid(ds2_cdiv(H, 4 * NRG), 2, NPART), block(NWP * 64);
    const int ncg = (ds2_cdiv(B, NPART) + 3) / 4;
    const size_t lds = (size_t)NRG * ncg * 16 * RED4_PITCH * sizeof(float);
#define DS2_BWD4_CASE(K)                                                                                         \
    case K:                                                                                                      \
        if (lds > 64 * 1024 &&                                                                                   \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_bwd_persistent4_kernel<K, NRG, PROTO, NP>),              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)             \
            return false;                                                                                        \
        if (!grid_is_coresident(&gru_bwd_persistent4_kernel<K, NRG, PROTO, NP>, grid, lds)) return false;                   \
        hipLaunchKernelGGL((gru_bwd_persistent4_kernel<K, NRG, PROTO, NP>), grid, block, lds, st, G, ghn, hout, d_out, w_hh_t,    \
                           sync, ring, T, B, H, dbg, spec_timing(1));                                                            \
        return true;
    switch (ngi) {
        DS2_BWD4_CASE(1)
        DS2_BWD4_CASE(2)
        DS2_BWD4_CASE(3)
        DS2_BWD4_CASE(5)
    }
#undef DS2_BWD4_CASE
    return false;
}

// the broadcast deal (gru_bwd_persistent5_kernel): three batch parts of one quad each, speculative hand-off
template <int NRG, int NPART = 3>
bool launch_bwd_persistent5(float* G, float* ghn, const float* hout, const float* d_out, const float* w_hh_t,
                            SyncWs* sync, float* ring, int T, int B, int H, int dbg, hipStream_t st) {
    const int opts[] = {1, 2, 3, 5};
    const int ngi = pick_kbw(ds2_cdiv(ds2_cdiv(3 * H, 64), NWP), opts, 4);
    dim3 grid(ds2_cdiv(H, 4 * NRG), 2, NPART), block(NWP * 64);
    // (first-attempt delay and adaptation policy of THIS kernel: DS2_GRU_BWD5_SPEC = "delay,inc,log2clean" for A/B runs)
    // (measured at B = 10, us per step, (14, 1, 2) -> (10, 1, 4): 28 units 2.95 -> 2.86, 24 units 2.74 -> 2.65; 20 units
    // 2.67-2.69 -> 2.54, and 2.51 with (10, 1, 5) -- with that the broadcast deal beats the 16-k-blocks deal's 2.61-2.64 there too)
    int d = NRG == 4 ? 8 : 10, inc = 1, l2c = NRG <= 5 ? 5 : 4;
    if (const char* e = ds2_tune_env("DS2_GRU_BWD5_SPEC")) sscanf(e, "%d,%d,%d", &d, &inc, &l2c);
    const int spec = spec_timing(1, d, inc, l2c);
#define DS2_BWD5_CASE(K)                                                                                         \
    case K:                                                                                                      \
        if (!grid_is_coresident(&gru_bwd_persistent5_kernel<K, NRG, NPART>, grid, 0)) return false;              \
        hipLaunchKernelGGL((gru_bwd_persistent5_kernel<K, NRG, NPART>), grid, block, 0, st, G, ghn, hout, d_out, w_hh_t, sync, ring, \
                           T, B, H, dbg, spec);                                                                  \
        return true;
    switch (ngi) {
        DS2_BWD5_CASE(1)
        DS2_BWD5_CASE(2)
        DS2_BWD5_CASE(3)
        DS2_BWD5_CASE(5)
    }
#undef DS2_BWD5_CASE
    return false;
}

// the d(h) hand-off (gru_bwd_persistent6_kernel): H = 32 RPW, one batch quad per part, speculative hand-off
template <int RPW, int NRG, int NPART = 3>
bool launch_bwd_persistent6(float* G, float* ghn, const float* hout, const float* d_out, const float* w_hh_t, const float* coef,
                            SyncWs* sync, float* ring, int T, int B, int H, int dbg, hipStream_t st) {
    dim3 grid(ds2_cdiv(H, 4 * NRG), 2, NPART), block(NWP * 64);
    auto kern = &gru_bwd_persistent6_kernel<RPW, NRG, NPART>;
#if defined(DS2_TIMING) || defined(DS2_FAULT_INJECT)
    // the ablated instantiations bench.py's floor leg and the timing tools ask for (the forms of a B = 9 .. 12 step at H = 800)
    if constexpr (RPW == 25 && NPART == 3 && (NRG == 5 || NRG == 7)) {
        const int abl = dbg & (2 | 2048 | 4096 | 8192);
        if (abl == (2 | 8192) || abl == 2) kern = &gru_bwd_persistent6_kernel<RPW, NRG, NPART, 2 | 8192>;
        else if (abl == (2048 | 8192) || abl == 2048) kern = &gru_bwd_persistent6_kernel<RPW, NRG, NPART, 2048 | 8192>;
        else if (abl == (4096 | 8192) || abl == 4096) kern = &gru_bwd_persistent6_kernel<RPW, NRG, NPART, 4096 | 8192>;
    }
#endif
    if (!grid_is_coresident(kern, grid, 0)) return false;
    // (first-attempt delay and adaptation policy of THIS kernel: DS2_GRU_BWD6_SPEC = "delay,inc,log2clean" for A/B runs)
    int d = 8, inc = 1, l2c = 5;
    if (const char* e = ds2_tune_env("DS2_GRU_BWD6_SPEC")) sscanf(e, "%d,%d,%d", &d, &inc, &l2c);
    hipLaunchKernelGGL(kern, grid, block, 0, st, G, ghn, hout, d_out, w_hh_t, coef, sync, ring, T, B, H, dbg,
                       spec_timing(1, d, inc, l2c));
    return true;
}

inline bool persistent_ok(int B, int H) {
    return (H % 16 == 0) && (2 * ds2_cdiv(H, PJU) <= max_persistent_wgs()) && (B <= 64) &&
           (ds2_cdiv(3 * H / 16, NWP) <= 19) && (ds2_cdiv(H / 16, NWP) <= 7);
}

// Hand-off protocol of the 4x4x1 forms: 0 counted + drained, 2 speculative loads (see CANARY_BITS).
// Default by the batch rows a workgroup handles: ONE batch quad (<= 4 rows: B <= 12 with three parts) -> speculative
// (B = 10: 3.45 -> 3.1 us per step, B = 8: 2.9 -> 2.5, B = 4: 2.5 -> 1.9); with more rows the step is MFMA / store bound, the
// canaries double the write-through stores and the drained protocol is faster (B = 32 backward: 6.4 vs 6.8 us, B = 64:
// 11.3 vs 12.8).  DS2_GRU_PROTO = 0 / 2 forces one (A/B timing).
inline int handoff_protocol(int rows_per_part) {
    const char* e = ds2_tune_env("DS2_GRU_PROTO");
    if (e && e[0] >= '0' && e[0] <= '2') return e[0] - '0';
    return rows_per_part <= 4 ? 2 : 0;
}

inline size_t header_bytes() { return ((sizeof(SyncWs) + 255) / 256) * 256; }

// diagnostics (results are WRONG when set): DS2_GRU_DBG bit0 = do not wait for arrivals, bit1 = skip the h loads +
// MFMAs, bit2 = skip the store drain, bit6 = workgroup 0 loses its arrival of step 2.  Compiled in only with
// -DDS2_TIMING=1 / -DDS2_FAULT_INJECT=1; the release library ignores the variable.
inline int dbg_flags() {
#if defined(DS2_TIMING) || defined(DS2_FAULT_INJECT)
    const char* e = getenv("DS2_GRU_DBG");
    return e ? atoi(e) : 0;
#else
    return 0;
#endif
}

}  // namespace

// A data-tagged-granule hand-off (Guideline 16 R2: every consumer wave polls the 8-byte {value, epoch} granules it
// needs) was built and measured in round 1: correct, but 1.5x (forward) to 2.1x (backward) SLOWER per step than the
// counter form -- 1600 waves polling payload lines swamp the fabric ("polling-cost" row of the price list).
// exchange ring: [dir 2][slot 2][batch tiles][k blocks of 16][16][16] floats, sized for the backward pass (K = 3H)
// DS2_GRU_P2_BF16 = 0 / 1: the two-part forms on the f32-input MFMA or on the bf16 pipe with split operands (default where
// H % 32 == 0 and a wave holds at most four forward k blocks of 32, i.e. H <= 1024)
inline bool p2_bf16(int H) {
    const char* e = getenv("DS2_GRU_P2_BF16");
    if (e && e[0] == '0') return false;
    return H % 32 == 0 && H <= 1024;
}
inline size_t ring_floats(int B, int H) {
    const int tiles16 = ds2_cdiv(B, 16) > 2 * ds2_cdiv(ds2_cdiv(B, 2), 16) ? ds2_cdiv(B, 16) : 2 * ds2_cdiv(ds2_cdiv(B, 2), 16);
    const size_t a = (size_t)tiles16 * (size_t)(3 * H / 16) * 256 * 3 / 2 + 4096;     // 16x16 forms (whole batch / two parts;
                                                                                      // three bf16 planes per value in the split forms)
    const size_t b2 = (size_t)2 * ds2_cdiv(B, 2), b3 = (size_t)3 * ds2_cdiv(B, 3);
    const size_t b = (b2 > b3 ? b2 : b3) * (size_t)ds2_cdiv(3 * H, 64) * 64;          // 4x4x1 forms (1-3 batch parts)
    return (size_t)2 * (2 * a > 4 * b ? 2 * a : 4 * b);                              // two slots / up to four (canary protocols)
}

extern "C" size_t ds2_gru_sync_ws_bytes(int B, int H) { return header_bytes() + ring_floats(B, H) * sizeof(float); }

extern "C" size_t ds2_gru_sync_error_offset(void) { return offsetof(SyncWs, error); }

extern "C" int ds2_gru_persistent_supported(int B, int H) { return persistent_ok(B, H) ? 1 : 0; }

extern "C" int ds2_gru_bwd_coef(const float* G, const float* ghn, const float* hout, float* coef, int T, int B, int H,
                                void* stream);
extern "C" int ds2_gru_bidir_fwd_persistent_ex(float* G, float* ghn, float* hout, const float* w_hh, float* coef,
                                               void* sync_ws, int T, int B, int H, void* stream);
extern "C" int ds2_gru_bidir_fwd_persistent(float* G, float* ghn, float* hout, const float* w_hh, void* sync_ws,
                                            int T, int B, int H, void* stream) {
    return ds2_gru_bidir_fwd_persistent_ex(G, ghn, hout, w_hh, nullptr, sync_ws, T, B, H, stream);
}

// coef != NULL: (T, B, 2, 3H), filled with the d(h)-hand-off backward recurrence's coefficient planes (see
// gru_bwd_persistent6_kernel) -- by the forward kernel's own gate threads where the form that runs has that output (the row
// deal: B = 9 .. 12 at H = 800), by an elementwise pass behind it on the same stream otherwise.
extern "C" int ds2_gru_bidir_fwd_persistent_ex(float* G, float* ghn, float* hout, const float* w_hh, float* coef,
                                               void* sync_ws, int T, int B, int H, void* stream) {
    DS2_CHECK_ARG(G && ghn && hout && w_hh && sync_ws);
    DS2_CHECK_ARG(T > 0 && B > 0 && H > 0);
    if (!persistent_ok(B, H)) {
        ds2_set_error("ds2_gru_bidir_fwd_persistent: unsupported shape B=%d H=%d", B, H);
        return DS2_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    SyncWs* sync = (SyncWs*)sync_ws;
    float* ring = (float*)((char*)sync_ws + header_bytes());
    // No memset: the previous launch left the counters zero (leave_kernel) and nothing reads ring bytes that this
    // launch has not written, except the padding batch columns of the 16x16x4 forms, whose products land in output
    // columns no gate thread reads.
    const int dbg = dbg_flags();
    bool ok;
    // Form by batch size.  Measured (H = 800, us per step, 4x4x1 broadcast form with the whole batch per workgroup vs
    // 16x16x4 form): B=4 2.57 / 3.65, B=10 3.84 / 3.88, B=16 4.41 / 3.89, B=32 7.44 / 5.56 -- the 4x4x1 form's cost
    // grows with every batch quad (and, with batch parts, with quads x parts for the MFMAs but only quads for the
    // loads), the 16x16x4 form's with every tile of 16.  DS2_GRU_FWD = "4" / "16" forces a form, DS2_GRU_FWD_SPLIT =
    // 1 / 2 / 3 the number of batch parts (A/B timing).
    const char* form = getenv("DS2_GRU_FWD");
    const bool ngi_ok = ds2_cdiv(ds2_cdiv(H, 64), NWP) <= 3;
    int parts = B <= 4 ? 1 : (B <= 8 ? 2 : 3);
    {
        const char* split = ds2_tune_env("DS2_GRU_FWD_SPLIT");
        if (split && split[0] >= '1' && split[0] <= '3') parts = split[0] - '0';
        if (parts > B) parts = B;
        // gate-role capacity: two passes of 512 / (4 * 8 parts) batch rows per workgroup; all workgroups co-resident
        while (parts < 3 && ds2_cdiv(B, parts) > 2 * (NWP * 64 / (32 * parts))) ++parts;
        if (2 * ds2_cdiv(H, 8 * parts) * parts > max_persistent_wgs()) parts = 1;
    }
    const int bper = ds2_cdiv(B, parts), rpp = NWP * 64 / (32 * parts);
    const bool fits4 = ngi_ok && bper <= 2 * rpp && (size_t)((bper + 3) / 4) * 4 * 24 * parts * FWD4_PITCH * 4 <= 140 * 1024;
    const bool use4 = (form ? form[0] == '4' : B <= 12) && fits4;
    if (use4) {
        const bool two = bper > rpp;
#define DS2_FWD4_GO(P_, N_)                                                                                         \
    (proto != 0 ? launch_fwd_persistent4<P_, N_, 2>(G, ghn, hout, w_hh, sync, ring, T, B, H, dbg, st)                \
                : launch_fwd_persistent4<P_, N_, 0>(G, ghn, hout, w_hh, sync, ring, T, B, H, dbg, st))
        const int proto = handoff_protocol(bper);
        // three parts, one batch quad each, speculative hand-off (B = 9 .. 12): 20-unit slices on 240 CUs when they fit
        const char* wide = getenv("DS2_GRU_FWD_WIDE");
        if (parts == 3 && !two && proto != 0 && (wide ? wide[0] == '1' : true) && H % 4 == 0 &&
            6 * ds2_cdiv(H, 20) <= max_persistent_wgs() && 6 * ds2_cdiv(H, 20) > 6 * ds2_cdiv(H, 24)) {
            // the row deal where it is built (H = 800: 100 columns per wave); DS2_GRU_FWD_ROWS = 0: the k-balanced deal (A/B timing)
            const char* rows = ds2_tune_env("DS2_GRU_FWD_ROWS");
            if (H == 800 && bper <= 4 && !(rows && rows[0] == '0'))
            {
                ok = launch_fwd_persistent5<100>(G, ghn, hout, w_hh, coef, sync, ring, T, B, H, dbg, st);
                if (ok) coef = nullptr;                   // written by the kernel itself
            }
            else
                ok = launch_fwd_persistent4<3, 1, 2, 5>(G, ghn, hout, w_hh, sync, ring, T, B, H, dbg, st);
        }
        // (the row deal with two batch parts -- 20 units, 160 workgroups -- measured at B = 8: 2.30 us per step against 2.15 for
        // the 16-unit k-balanced form on 200: its 48 gate rows fill 12 of the 16 blocks, the row deal would issue 100 MFMAs for 75)
        else
        if (parts == 1) ok = two ? DS2_FWD4_GO(1, 2) : DS2_FWD4_GO(1, 1);
        else if (parts == 2) ok = two ? DS2_FWD4_GO(2, 2) : DS2_FWD4_GO(2, 1);
        else ok = two ? DS2_FWD4_GO(3, 2) : DS2_FWD4_GO(3, 1);
#undef DS2_FWD4_GO
    }
    // (round 4, us per step at H = 800: the split-operand two-part form costs 3.32-3.36 whatever B <= 32 is; the whole-batch
    // 16x16x4 form 3.63 at B = 16 -- so the split form takes over as soon as the 4x4x1 forms end, at B = 13)
    else if ((ds2_tune_env("DS2_GRU_FWD_P2") ? ds2_tune_env("DS2_GRU_FWD_P2")[0] == '1' : B >= (p2_bf16(H) ? 13 : 17)) && B >= 2 && H % 16 == 0)
        ok = p2_bf16(H) ? ((B + 1) / 2 <= 16 ? ds2_p16_launch_fwd(2, 1, G, ghn, hout, w_hh, sync, ring, T, B, H, dbg, st)
                                             : ds2_p16_launch_fwd(2, 2, G, ghn, hout, w_hh, sync, ring, T, B, H, dbg, st))
                        : ((B + 1) / 2 <= 16 ? ds2_p16_launch_fwd(1, 1, G, ghn, hout, w_hh, sync, ring, T, B, H, dbg, st)
                                             : ds2_p16_launch_fwd(1, 2, G, ghn, hout, w_hh, sync, ring, T, B, H, dbg, st));
    else if (B <= 16) ok = ds2_p16_launch_fwd(0, 1, G, ghn, hout, w_hh, sync, ring, T, B, H, dbg, st);
    else if (B <= 32) ok = ds2_p16_launch_fwd(0, 2, G, ghn, hout, w_hh, sync, ring, T, B, H, dbg, st);
    else ok = ds2_p16_launch_fwd(0, 4, G, ghn, hout, w_hh, sync, ring, T, B, H, dbg, st);
    if (!ok) {
        ds2_set_error("ds2_gru_bidir_fwd_persistent: the chosen kernel's grid is not co-resident on this device (B=%d H=%d)", B, H);
        return DS2_ERR_UNSUPPORTED;
    }
    DS2_CHECK_LAUNCH();
    if (coef) return ds2_gru_bwd_coef(G, ghn, hout, coef, T, B, H, stream);
    return DS2_OK;
}

// spare_cus: how many compute units the caller wants LEFT FREE beside this launch (work it has queued on other streams: the
// weight-gradient GEMMs of the layer above, a collective); < 0 = the library's default.  Where the 4x4x1 forms run three batch
// parts with the speculative hand-off (B = 9 .. 12) a workgroup may own 20, 24 or 28 units, i.e. 240, 204 or 174 workgroups at
// H = 800 -- measured at B = 10, us per step stand-alone: 2.65 / 2.82 / 3.0 -- and inside a training step the SLOWER forms win
// wherever the side stream has GEMMs waiting (they only get the CUs a recurrence launch leaves free): the launch picks the
// widest grid that leaves spare_cus free.  Everywhere else the hint changes nothing -- measured for the two-part forms (B = 5 .. 8:
// 16 / 20 / 24 units per workgroup = 200 / 160 / 136 workgroups, 2.34 / 2.56 / 2.79 us per step at B = 8): there the narrower
// grids LOSE in the whole step (B = 8 x 15 s: 23.2 / 23.6 / 24.1 ms), the recurrence pays more than the hidden GEMMs return.
extern "C" int ds2_gru_bidir_bwd_persistent_ex(float* G, float* ghn, const float* hout, const float* d_out,
                                               const float* w_hh_t, void* sync_ws, int T, int B, int H, int spare_cus,
                                               void* stream);

extern "C" int ds2_gru_bidir_bwd_persistent(float* G, float* ghn, const float* hout, const float* d_out,
                                            const float* w_hh_t, void* sync_ws, int T, int B, int H, void* stream) {
    return ds2_gru_bidir_bwd_persistent_ex(G, ghn, hout, d_out, w_hh_t, sync_ws, T, B, H, -1, stream);
}

extern "C" int ds2_gru_bidir_bwd_persistent_ex(float* G, float* ghn, const float* hout, const float* d_out,
                                               const float* w_hh_t, void* sync_ws, int T, int B, int H, int spare_cus,
                                               void* stream) {
    DS2_CHECK_ARG(G && ghn && hout && d_out && w_hh_t && sync_ws);
    DS2_CHECK_ARG(T > 0 && B > 0 && H > 0);
    if (!persistent_ok(B, H)) {
        ds2_set_error("ds2_gru_bidir_bwd_persistent: unsupported shape B=%d H=%d", B, H);
        return DS2_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    SyncWs* sync = (SyncWs*)sync_ws;
    float* ring = (float*)((char*)sync_ws + header_bytes());
    const int dbg = dbg_flags();
    bool ok;
    // Form: the 4x4x1 forms up to B = 16; from B = 17 the two-part 16x16x4 form (us per step at H = 800, B = 32 / 64: 4x4x1
    // 7.0 / 12.3, see gru_bwd_persistent_p2_kernel).  DS2_GRU_BWD = "4" / "16" forces a family, DS2_GRU_BWD_P2 = 0 / 1 the
    // two-part 16x16x4 form off / on (A/B timing).
    const char* form = getenv("DS2_GRU_BWD");
    const char* p2e = ds2_tune_env("DS2_GRU_BWD_P2");
    const bool p2 = (p2e ? p2e[0] == '1' : (B >= 17 && !(form && form[0] == '4'))) && B >= 2 && H % 16 == 0;
    const bool use4 = !p2 && (form ? form[0] == '4' : true);
    // Batch parts (1, 2 or 3: 8, 16 or 24 units per workgroup) by the fitted cost model; measured, H = 800, us per step,
    // whole batch / two halves: B=8 3.32 / 2.98, B=10 3.90 / 4.04, B=16 4.77 / 4.09, B=32 7.48 / 6.51, B=64 12.84 / 11.38.
    // DS2_GRU_BWD_SPLIT = 1 / 2 / 3 forces a form (A/B timing).
    int parts = 1;
    {
        double best = 1e30;
        for (int p = 1; p <= 3; ++p) {
            if (p > B || 2 * ds2_cdiv(H, 8 * p) * p > max_persistent_wgs() || ds2_cdiv(B, p) * 8 * p > NWP * 64) continue;
            const int bper = ds2_cdiv(B, p), quads = (bper + 3) / 4;
            const double cost = 0.40 * quads * p + 0.34 * bper / 4.0 + (p == 3 && quads > 2 ? 0.3 : 0.0);
            if (cost < best - 1e-9) {
                best = cost;
                parts = p;
            }
        }
        const char* split = ds2_tune_env("DS2_GRU_BWD_SPLIT");
        if (split && split[0] >= '1' && split[0] <= '3' && (split[0] - '0') <= B) parts = split[0] - '0';
    }
    const bool ngi_ok = ds2_cdiv(ds2_cdiv(3 * H, 64), NWP) <= 5;
    const int proto = handoff_protocol(ds2_cdiv(B, parts));
#define DS2_BWD4_GO(R_)                                                                                             \
    (proto != 0 ? launch_bwd_persistent4<R_, 2>(G, ghn, hout, d_out, w_hh_t, sync, ring, T, B, H, dbg, st)           \
                : launch_bwd_persistent4<R_, 0>(G, ghn, hout, d_out, w_hh_t, sync, ring, T, B, H, dbg, st))
    // (round 5, measured and removed: 17 <= B <= 32 on the 4x4x1 instruction with 32 units and a QUARTER of the batch per
    // workgroup -- half the hand-off bytes of the two-part form per multiply-add, counted hand-off -- B = 32 / 24 / 17: 5.00 /
    // 4.70 / 4.52 us per step against 4.74 / 4.58 / 4.48: the 4x4x1 instruction retires a multiply-add in 10 cycles / 256 where
    // the 16x16x4 one takes 32 / 1024, and the bytes are not what bounds the two-part form: from B = 17 to 32 its part grows
    // from 9 to 16 rows, 84 to 150 KB per workgroup and step, and the step by 0.26 us -- profiles/r05_recurrence_experiments.md)
    if (p2)
        // (the backward twin is hand-off-bound, not matrix-bound -- three gates' values cross per step, and as bf16 planes
        // they are 1.5 x the bytes: 4.91 against 4.74 us per step at B = 32, 8.46 against 7.57 at B = 64; DS2_GRU_P2_BF16_BWD=1
        // selects it for A/B runs)
        ok = (p2_bf16(H) && getenv("DS2_GRU_P2_BF16_BWD") && getenv("DS2_GRU_P2_BF16_BWD")[0] == '1')
                 ? ((B + 1) / 2 <= 16 ? ds2_p16_launch_bwd(2, 1, G, ghn, hout, d_out, w_hh_t, sync, ring, T, B, H, dbg, st)
                                      : ds2_p16_launch_bwd(2, 2, G, ghn, hout, d_out, w_hh_t, sync, ring, T, B, H, dbg, st))
                 : ((B + 1) / 2 <= 16 ? ds2_p16_launch_bwd(1, 1, G, ghn, hout, d_out, w_hh_t, sync, ring, T, B, H, dbg, st)
                                      : ds2_p16_launch_bwd(1, 2, G, ghn, hout, d_out, w_hh_t, sync, ring, T, B, H, dbg, st));
    else if (use4 && ngi_ok && parts == 3 && proto != 0 && ds2_cdiv(B, 3) <= 4) {   // (<= 4: also under a forced DS2_GRU_PROTO)
        // units per workgroup by the CUs to leave free (see above); DS2_GRU_BWD_WIDE = 0 / 1 / 2 forces 24 / 28 / 20 (A/B timing)
        const int cus = device_cus();
        int want = spare_cus < 0 ? 52 : spare_cus;
        const char* w = ds2_tune_env("DS2_GRU_BWD_WIDE");
        if (w && w[0] >= '0' && w[0] <= '2') want = w[0] == '0' ? 52 : (w[0] == '1' ? 82 : 0);
        const int g20 = 6 * ds2_cdiv(H, 20), g24 = 6 * ds2_cdiv(H, 24), g28 = 6 * ds2_cdiv(H, 28);
        // the broadcast deal (gru_bwd_persistent5_kernel): H is a multiple of 16 and a part is one batch quad here.  Measured
        // stand-alone at B = 10, us per step, against the 16-k-blocks deal it replaces (gru_bwd_persistent4_kernel<5, NRG, 2, 3>,
        // removed in round 5), each with its own hand-off timing: 28 units 3.02-3.05 -> 2.86, 24 units 2.79-2.81 -> 2.65,
        // 20 units 2.61-2.64 -> 2.51-2.54
        if (g20 <= max_persistent_wgs() && g20 > g24 && cus - g20 >= want)
            ok = launch_bwd_persistent5<5>(G, ghn, hout, d_out, w_hh_t, sync, ring, T, B, H, dbg, st);         // 240 workgroups
        else if (cus - g24 >= want || g28 >= g24)
            ok = launch_bwd_persistent5<6>(G, ghn, hout, d_out, w_hh_t, sync, ring, T, B, H, dbg, st);         // 204
        else
            ok = launch_bwd_persistent5<7>(G, ghn, hout, d_out, w_hh_t, sync, ring, T, B, H, dbg, st);         // 174
    }
    else if (use4 && ngi_ok && parts == 3)           // (three parts of more than one quad: the counted protocol)
        ok = launch_bwd_persistent4<6, 0>(G, ghn, hout, d_out, w_hh_t, sync, ring, T, B, H, dbg, st);
    else if (use4 && ngi_ok && parts == 2 && proto != 0 && ds2_cdiv(B, 2) <= 4)
        // B = 5 .. 8: 16 units and half the batch per workgroup in the broadcast deal (set A only: no fold at all).  us per
        // step against the 16-k-blocks deal (gru_bwd_persistent4_kernel<5, 4, 2>), B = 8 / 6 / 5: 2.34 / 2.30 / 2.30 -> 2.20 /
        // 2.15 / 2.11 -- with round 2's hand-off timing the two were level (2.33-2.37 against 2.34-2.35 at B = 8)
        ok = launch_bwd_persistent5<4, 2>(G, ghn, hout, d_out, w_hh_t, sync, ring, T, B, H, dbg, st);
    else if (use4 && ngi_ok && parts == 2) ok = DS2_BWD4_GO(4);
    else if (use4 && ngi_ok) ok = DS2_BWD4_GO(2);
    else if (B <= 16) ok = ds2_p16_launch_bwd(0, 1, G, ghn, hout, d_out, w_hh_t, sync, ring, T, B, H, dbg, st);
    else if (B <= 32) ok = ds2_p16_launch_bwd(0, 2, G, ghn, hout, d_out, w_hh_t, sync, ring, T, B, H, dbg, st);
    else ok = ds2_p16_launch_bwd(0, 4, G, ghn, hout, d_out, w_hh_t, sync, ring, T, B, H, dbg, st);
    if (!ok) {
        ds2_set_error("ds2_gru_bidir_bwd_persistent: the chosen kernel's grid is not co-resident on this device (B=%d H=%d)", B, H);
        return DS2_ERR_UNSUPPORTED;
    }
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

// ---- the d(h) hand-off backward recurrence (ABI revision 402) ---------------------------------------------------------------
// Shapes it is built for: H = 800 (and H = 64, the test width), 5 <= B <= 12 -- the speculative one-quad-per-part forms.
static bool dh_form_ok(int B, int H) {
    return (H == 800 || H == 64) && B >= 5 && B <= 12 && persistent_ok(B, H);
}
extern "C" int ds2_gru_bwd_dh_supported(int B, int H) { return dh_form_ok(B, H) ? 1 : 0; }

extern "C" int ds2_gru_bwd_coef(const float* G, const float* ghn, const float* hout, float* coef, int T, int B, int H,
                                void* stream) {
    DS2_CHECK_ARG(G && ghn && hout && coef);
    DS2_CHECK_ARG(T > 0 && B > 0 && H > 0 && H % 4 == 0);
    const size_t n4 = (size_t)T * B * 2 * (H / 4);
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(gru_bwd_coef_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, G, ghn, hout, coef, T, B, H);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" int ds2_gru_bidir_bwd_persistent_dh(float* G, float* ghn, const float* hout, const float* d_out,
                                               const float* w_hh_t, const float* coef, void* sync_ws, int T, int B, int H,
                                               int spare_cus, void* stream) {
    DS2_CHECK_ARG(G && ghn && hout && d_out && w_hh_t && coef && sync_ws);
    DS2_CHECK_ARG(T > 0 && B > 0 && H > 0);
    if (!dh_form_ok(B, H)) {
        ds2_set_error("ds2_gru_bidir_bwd_persistent_dh: unsupported shape B=%d H=%d", B, H);
        return DS2_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    SyncWs* sync = (SyncWs*)sync_ws;
    float* ring = (float*)((char*)sync_ws + header_bytes());
    const int dbg = dbg_flags();
    bool ok = false;
#define DS2_BWD6_GO(RPW_, NRG_, NP_) launch_bwd_persistent6<RPW_, NRG_, NP_>(G, ghn, hout, d_out, w_hh_t, coef, sync, ring, T, B, H, dbg, st)
    if (B >= 9) {                                       // three parts of one batch quad: units per workgroup by the CUs to leave free
        const int cus = device_cus();
        int want = spare_cus < 0 ? 52 : spare_cus;
        const char* w = ds2_tune_env("DS2_GRU_BWD_WIDE");
        if (w && w[0] >= '0' && w[0] <= '2') want = w[0] == '0' ? 52 : (w[0] == '1' ? 82 : 0);
        const int g20 = 6 * ds2_cdiv(H, 20), g24 = 6 * ds2_cdiv(H, 24), g28 = 6 * ds2_cdiv(H, 28);
        if (H == 800) {
            if (g20 <= max_persistent_wgs() && g20 > g24 && cus - g20 >= want) ok = DS2_BWD6_GO(25, 5, 3);
            else if (cus - g24 >= want || g28 >= g24) ok = DS2_BWD6_GO(25, 6, 3);
            else ok = DS2_BWD6_GO(25, 7, 3);
        } else {
            ok = want >= 80 ? DS2_BWD6_GO(2, 7, 3) : (want >= 40 ? DS2_BWD6_GO(2, 6, 3) : DS2_BWD6_GO(2, 5, 3));
        }
    } else {                                            // B = 5 .. 8: two parts, 16 units
        ok = H == 800 ? DS2_BWD6_GO(25, 4, 2) : DS2_BWD6_GO(2, 4, 2);
    }
#undef DS2_BWD6_GO
    if (!ok) {
        ds2_set_error("ds2_gru_bidir_bwd_persistent_dh: the chosen kernel's grid is not co-resident on this device (B=%d H=%d)", B, H);
        return DS2_ERR_UNSUPPORTED;
    }
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

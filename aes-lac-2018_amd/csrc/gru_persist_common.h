// Shared by the persistent BiGRU recurrence translation units (gru_persist.hip: the 4x4x1 forms and the C entry points;
// gru_persist16.hip: the 16x16 forms).  Everything here sits in an anonymous namespace: each unit gets its own copy.
#pragma once
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <type_traits>

#include "ds2_common.h"

// per-phase / per-wave stamps exist in the unit that defines the buffers (gru_persist.hip, -DDS2_TIMING=1) only
#if !defined(DS2_PERSIST_MAIN_TU)
#define DS2_TICK(i) do {} while (0)
#define DS2_WTICK(i) do {} while (0)
#define DS2_RETRY_FLUSH(n) do {} while (0)
#endif

namespace {


constexpr int NWP = 8;                    // waves per persistent workgroup (2 per SIMD)
constexpr int PJU = 8;                    // hidden units per workgroup
#ifdef DS2_FAULT_INJECT
constexpr unsigned long long SPIN_TICKS = 150000000ull;  // (the fault-injection build, whose tests WAIT for a time-out: 1.5 s)
#else
constexpr unsigned long long SPIN_TICKS = 500000000ull;  // 5 s of the 100 MHz real-time counter: long enough to
                                                         // sit out a peer workgroup that is waiting for CUs held by
                                                         // a concurrent RCCL kernel, short enough to end a lost run
#endif

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int NSHARD = 8;  // arrival counters per direction (workgroup x -> shard x % 8), each on its own 128-B line:
                           // 100 arrivals on ONE word serialise at ~12 ns each (MI355X_MICROARCH.md "fanin")
struct SyncWs {            // lives in caller-provided device memory: zeroed ONCE by the caller when it is allocated (and
                           // again after a reported timeout); every launch that completes leaves the counters zero
    unsigned int arrive[2][3][NSHARD][32];   // [direction][batch part (backward batch-split forms)][shard][line]
    unsigned int done[32];   // workgroups that have left the kernel; the last one zeroes arrive[] and done for the next launch
    unsigned int error;      // set to 1 on a spin timeout; STICKY: only the host clears it (ops.raise_async_error)
};

// Diagnostics that change results (skip the arrival wait / the MFMAs / the store drain, lose an arrival) exist only in
// builds made with -DDS2_TIMING=1 or -DDS2_FAULT_INJECT=1 (tools/gru_*_timing.py, the lost-arrival test's library):
// the release library never reads DS2_GRU_DBG.
#if defined(DS2_TIMING) || defined(DS2_FAULT_INJECT)
#define DS2_DBG(dbg, bit) ((dbg) & (bit))
#else
#define DS2_DBG(dbg, bit) 0
#endif

// s_waitcnt vmcnt(0) as the BUILTIN (simm16: vmcnt = 0, expcnt and lgkmcnt at their maxima = not waited for), not as inline
// asm: the compiler's wait-count insertion cannot see into an asm statement, so behind one it still believes the wave's
// earlier loads are in flight -- and when a later instruction touches one of their destination registers (the loop-carried
// saved-activation registers of the 4x4x1 kernels) it inserts a vmcnt(0) of its OWN at that point, which then also waits for
// whatever was issued in between: round 4 found such a wait right behind the hand-off stores (0.4 us per step waiting for
// the write-through acknowledgements before the next loads could issue), and, in the round-3 kernels, at the top of every step.
__device__ __forceinline__ void wait_vmcnt0() {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
}
// Called by thread 0 of every workgroup that leaves the kernel normally (not on the timeout path: there the host resets
// the workspace).  The arrival adds of this workgroup have been performed at the memory side once vmcnt is 0; the
// workgroup whose add to `done` comes last knows every other workgroup has stopped polling and adding, and zeroes the
// counters with write-through stores: the next launch on the stream starts from zero without a memset in between.
__device__ __forceinline__ void leave_kernel(SyncWs* sync) {
    wait_vmcnt0();
    const unsigned int total = gridDim.x * gridDim.y * gridDim.z;
    const unsigned int prev = __hip_atomic_fetch_add(&sync->done[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == total - 1) {
        unsigned int* a = &sync->arrive[0][0][0][0];
        for (int i = 0; i < 2 * 3 * NSHARD; ++i) __hip_atomic_store(a + i * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sync->done[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The scalar offset is an OPAQUE zero (an s_mov the optimiser cannot see through).  Without it two hand-off loads of one
// address are, to the compiler, the same value: the exchange ring is reached through a __restrict__ kernel argument, so not
// even an asm "memory" barrier between them says that somebody else may have written it -- and the speculative protocol's
// RE-load of a fragment that still held the canary was folded into the first load's result in one instantiation
// (gru_bwd_persistent6_kernel<2, ..>, round 6: the retry loop spun on a stale register until the time-out; the instantiations
// that shipped in rounds 2-5 happened to keep their re-loads -- their retry counters say so -- but nothing guaranteed it).
__device__ __forceinline__ f32x4 load_sc1_b128(__amdgpu_buffer_rsrc_t rsrc, int byte_off) {
    int zero = 0;
    asm volatile("" : "+s"(zero));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, zero, 16 /* sc1 */);
    return __builtin_bit_cast(f32x4, v);
}
// hand-off payload load (a timing experiment with plain, L2-allocating loads showed no difference; a run-time
// switch between the two forms here costs a branch and a vmcnt(0) join in front of the MFMAs)
#define LOAD_HANDOFF(rs, off) load_sc1_b128(rs, off)
constexpr int OOB_OFFSET = 0x7FFFFFF0;    // beyond any descriptor's num_records: the range-checked load returns 0
__device__ __forceinline__ void store_sc1(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Gate nonlinearities on the hardware exp2 / rcp (1 ulp each): the gate math sits on the step's critical path, and the
// library expf / tanhf / IEEE division cost ~60 more instructions there.  |error| < 3e-7, saturates correctly.
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// Speculative hand-off (4x4x1 forms, PROTO = 2; chosen when a workgroup handles one batch quad, i.e. B <= 12).
// The counted protocol's step is a chain of four dependent memory operations -- payload stores -> their acks (the drain,
// ~0.45 us) -> arrival add -> (others) poll -> payload loads -- about 2 us of every 3.45 us step at B = 10.  Here a step has
// TWO: payload stores, payload loads.  There are no per-step counters, no drain before a signal and no poll: a consumer simply
// loads the slot and checks that what it got is payload.  To make that check possible every ring position is overwritten
// with CANARY_BITS (a NaN pattern arithmetic never produces; payload values that alias it are re-encoded) TWO steps before
// its next payload, by the same lane (same address: the stores stay ordered), in a ring of FOUR slots; a wave that finds the
// pattern in a fragment loads that fragment again (bounded by SPIN_TICKS).  Why a stale payload can never pass for a new
// one: a workgroup waits for its previous step's stores (a step old, so the wait is free) before it issues a step's
// payload, so a consumer that has seen producer P's payload of step s knows P's canaries of steps < s have landed -- and it
// must have seen that payload to finish its own step s + 1, before it reads the slot of step s + 1, whose canary P wrote in
// step s - 1.  One counted rendezvous per launch covers the start (slots 0 and 1 canaried by everyone before anyone reads).
// A wave times its first attempt (an adaptive s_sleep count, +1 after a step with a re-load, -1 after four clean ones): a
// failed attempt costs a round trip and, from 1600 waves, polling traffic.  Measured (H = 800, us per step fwd / bwd):
// B = 10: 3.48 / 3.44 -> 3.1 / 3.1; B = 8: 2.90 / 2.93 -> 2.45 / 2.55; B = 4: 2.52 / 2.52 -> 1.9 / 2.1.
// (A "signal first, drain later" variant -- canaries one slot ahead, counters kept, the drain moved behind the arrival add --
// gained 3 %; it is gone.)
constexpr unsigned int CANARY_BITS = 0xFFFFFFFFu;
__device__ __forceinline__ void store_canary(float* p) {
    __hip_atomic_store(reinterpret_cast<unsigned int*>(p), CANARY_BITS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one 16-byte write-through (sc1) store through a WAVE-UNIFORM descriptor + per-lane byte offset (a per-lane descriptor
// would make the compiler emit a waterfall loop over the lanes)
__device__ __forceinline__ void store_sc1_b128(__amdgpu_buffer_rsrc_t rs, int byte_off, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, byte_off, 0, 16 /* sc1 */);
}
// a payload value must never look like the canary: a NaN with the all-ones payload (only reachable from NaN inputs carrying
// that payload) is re-encoded as the canonical quiet NaN -- still a NaN for every consumer
__device__ __forceinline__ float not_canary(float v) {
    return __float_as_uint(v) == CANARY_BITS ? __uint_as_float(0x7FC00000u) : v;
}
__device__ __forceinline__ bool has_canary(f32x4 v) {
    const u32x4 u = __builtin_bit_cast(u32x4, v);
    return (u[0] == CANARY_BITS) | (u[1] == CANARY_BITS) | (u[2] == CANARY_BITS) | (u[3] == CANARY_BITS);
}
// compile-time loop: f(std::integral_constant<int, I>{}) for I = 0 .. N - 1 (the MFMA's ABID operand must be an immediate)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// canary test on the components MASK names (bit e = component e): the k-balanced deal's padding components are never written
template <int MASK>
__device__ __forceinline__ bool has_canary_masked(f32x4 v) {
    const u32x4 u = __builtin_bit_cast(u32x4, v);
    bool r = false;
    if (MASK & 1) r |= u[0] == CANARY_BITS;
    if (MASK & 2) r |= u[1] == CANARY_BITS;
    if (MASK & 4) r |= u[2] == CANARY_BITS;
    if (MASK & 8) r |= u[3] == CANARY_BITS;
    return r;
}


// Speculative protocol: make sure no loaded fragment still holds the canary (returns true if one did).  Runs before the
// MFMAs; stale fragments are re-loaded per load instruction (a wave-uniform decision) until clean, bounded by SPIN_TICKS like
// every other spin.  Also adapts the wave's first-attempt delay: +1 after a step that needed a re-load, -1 after 4 clean steps.
// NVC > 0 (the k-balanced forward deal): only the first NVC components of a lane's NGI loads are ever written (component
// c of load gi is number 4 gi + c); the others are padding nobody owns and must not be mistaken for a missing payload.
// G0 .. G1 - 1: the loads this call checks (STAGED consumption, round 4: a k group's fragments are validated right before
// its own MFMAs, so that the matrix work on the fragments that have landed runs under the wait for the rest -- the chip-wide
// all-to-all delivers a step's hand-off loads over ~0.8 us, first issued first served: tools/gru_wave_timing.py); `acc`
// carries "a re-load happened" from stage to stage and the delay adapts once, in the call with `last` set.
template <int NCI, int NGI, int NVC = 0, int G0 = 0, int G1 = NGI, typename LoadFrag>
__device__ __forceinline__ bool validate_fragments(f32x4 (&bf)[NCI][NGI], LoadFrag& load_frag, bool first_chunk, int spec,
                                                   int& spec_delay, int& spec_clean, SyncWs* sync, int& abort_flag,
                                                   int& nretry, bool* acc = nullptr, bool last = true) {
    bool retried = false;
    unsigned long long t_retry = 0;
    auto has_canary = [](f32x4 v, auto gi_tag) {
        constexpr int gi = decltype(gi_tag)::value;
        constexpr int left = NVC > 0 ? NVC - 4 * gi : 4;
        return has_canary_masked<(left >= 4 ? 15 : (1 << (left > 0 ? left : 0)) - 1)>(v);
    };
    if constexpr (NCI * (G1 - G0) <= 2) {   // the common case -- every fragment already holds payload -- costs ONE ballot (with the
                                      // backward kernel's five fragments the joint test measured 0.05 us per step SLOWER)
        bool stale = false;
        static_for<G0, G1>([&](auto gi_tag) {
#pragma unroll
            for (int ci = 0; ci < NCI; ++ci) stale |= has_canary(bf[ci][decltype(gi_tag)::value], gi_tag);
        });
        if (!__any(stale)) goto validated;
    }
    for (;;) {
        asm volatile("" ::: "memory");                            // (keeps re-loads from being hoisted or merged)
        bool any = false;
        static_for<G0, G1>([&](auto gi_tag) {
            constexpr int gi = decltype(gi_tag)::value;
#pragma unroll
            for (int ci = 0; ci < NCI; ++ci)
                if (__any(has_canary(bf[ci][gi], gi_tag))) {
                    any = true;
                    load_frag(ci, gi);
                }
        });
        if (!any) break;
        ++nretry;
        retried = true;
        for (int i = 0; i < ((spec >> 8) & 0xFF); ++i) __builtin_amdgcn_s_sleep(1);
        if (t_retry == 0) t_retry = __builtin_amdgcn_s_memrealtime();
        if (__builtin_amdgcn_s_memrealtime() - t_retry > SPIN_TICKS) {   // a payload never came: flag it, leave the launch
            if ((threadIdx.x & 63) == 0) {
                __hip_atomic_store(&sync->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                abort_flag = 1;
            }
            break;
        }
    }
    // Round 6: behind the re-load loop NOTHING is in flight.  The loop's exits are decided by scalar masks the compiler's
    // wait-count pass cannot follow, so it merged "a re-issued load is this wave's YOUNGEST operation" into the join with the
    // one-ballot fast path above and put a vmcnt(0) in front of the first MFMA of EVERY step -- the later fragments' staged
    // consumption never happened on the fast path either (seen in the ISA of gru_fwd_persistent5_kernel: vmcnt(1), ballot,
    // then vmcnt(0) before the first of the 64 MFMAs).  With this wait on the slow path only, the join needs none.
    if constexpr (NCI * (G1 - G0) <= 2) wait_vmcnt0();
validated:
    if (acc) {
        *acc |= retried;
        retried = *acc;
    }
    if (!last) return retried;
    if (((spec >> 16) & 1) && first_chunk) {
        if (retried) {
            spec_delay = min(spec_delay + ((spec >> 18) & 3), 63);
            spec_clean = 0;
        } else if (++spec_clean == (1 << ((spec >> 20) & 7))) {
            spec_clean = 0;
            spec_delay = max(spec_delay - 1, 0);
        }
    }
    // wave-uniform by construction (every decision above is a ballot): tell the compiler, so that the first-attempt sleep
    // and this bookkeeping are scalar code instead of exec-masked vector loops
    spec_delay = __builtin_amdgcn_readfirstlane(spec_delay);
    spec_clean = __builtin_amdgcn_readfirstlane(spec_clean);
    return retried;
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

inline int pick_kbw(int need, const int* opts, int nopts) {
    for (int i = 0; i < nopts; ++i)
        if (opts[i] >= need) return opts[i];
    return -1;
}

// Co-residency: every workgroup of a persistent launch spins on arrivals from all the others, so the whole grid must be
// on the chip at once.  The budget is 15/16 of the CURRENT device's compute units (240 of an MI355X's 256: the rest
// stays free for a concurrent RCCL kernel or the side stream), read once per device; a partitioned (CPX) or smaller
// device simply answers "unsupported" and the caller uses the per-step kernels.
inline int device_cus() {
    static int cus[64] = {0};                          // immutable once filled; a race writes the same value twice
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cus[dev] = n;
    }
    return cus[dev];
}
inline int max_persistent_wgs() {
    const int n = device_cus();
    return n - n / 16;
}

// Launch-time check of the kernel actually chosen: blocks per CU for its register / LDS footprint x CUs >= grid.
// Answers are cached per (kernel, LDS bytes, device) -- the occupancy query costs tens of microseconds.
template <typename K>
inline bool grid_is_coresident(K kernel, dim3 grid, size_t lds) {
    struct Entry { const void* k; size_t lds; int dev; int per_cu; };
    static Entry cache[64];
    static int ncache = 0;
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const void* key = reinterpret_cast<const void*>(kernel);
    int per_cu = -1;
    {
        std::lock_guard<std::mutex> lock(mu);
        for (int i = 0; i < ncache; ++i)
            if (cache[i].k == key && cache[i].lds == lds && cache[i].dev == dev) per_cu = cache[i].per_cu;
        if (per_cu < 0) {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, NWP * 64, lds) != hipSuccess) n = 0;
            per_cu = n;
            if (ncache < 64) cache[ncache++] = Entry{key, lds, dev, n};
        }
    }
    return (long)grid.x * grid.y * grid.z <= (long)per_cu * device_cus();
}

}  // namespace

// the 16x16 forms (gru_persist16.hip), launched by the entry points of gru_persist.hip: form 0 = whole batch per workgroup,
// 1 = two batch parts (f32-input MFMA), 2 = two batch parts on the bf16 pipe with split operands; nbt = batch tiles of 16 per
// workgroup (1, 2 or 4 for form 0; 1 or 2 otherwise).  false: the grid is not co-resident (or no such instantiation)
bool ds2_p16_launch_fwd(int form, int nbt, float* G, float* ghn, float* hout, const float* w_hh, void* sync, float* ring, int T,
                        int B, int H, int dbg, hipStream_t st);
bool ds2_p16_launch_bwd(int form, int nbt, float* G, float* ghn, const float* hout, const float* d_out, const float* w_hh_t,
                        void* sync, float* ring, int T, int B, int H, int dbg, hipStream_t st);

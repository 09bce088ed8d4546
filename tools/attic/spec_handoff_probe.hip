// Probe: per-step cost of a TWO-operation all-to-all hand-off (stores -> loads, the payload is its own tag: no counters, no
// drain) among the 32 workgroups of a group, when the group is (a) the workgroups that find themselves on one XCD, with plain
// stores (kept in that XCD's L2) or sc1 (write-through) stores, and (b) 32 consecutive blocks (spread over all XCDs), sc1 stores.
// 256 workgroups of 512 threads (1 per CU); NG groups run, the rest exit.  "work" = dependent v_fma chain (cycles) between a
// step's loads and its stores, standing in for the MFMA + gate phase.
//   hipcc --offload-arch=gfx950 -O2 tools/spec_handoff_probe.hip -o tools/spec_handoff_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Ctl {
    unsigned int members[8][32];
    unsigned int total[32];
    unsigned int retries;
    unsigned int census[8];
};

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xF; }

// SMODE 0 plain stores, 1 sc1 stores; LAUX: aux bits of the loads (16 sc1, 2 nt, 17 sc0 sc1)
template <int SMODE, int LAUX>
__global__ __launch_bounds__(512) void probe(Ctl* ctl, float* ring, int steps, int floats_per_wg, int by_block, int ngroups,
                                             int work, int delay) {
    __shared__ unsigned s_group, s_member;
    const int tid = threadIdx.x;
    if (tid == 0) {
        unsigned g, m;
        if (by_block) {
            g = blockIdx.x >> 5;
            m = blockIdx.x & 31;
        } else {
            g = xcc_id();
            m = __hip_atomic_fetch_add(&ctl->members[g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __hip_atomic_fetch_add(&ctl->total[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(&ctl->total[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) __builtin_amdgcn_s_sleep(4);
        s_group = g;
        s_member = m;
        if (m == 0 && !by_block) ctl->census[g] = __hip_atomic_load(&ctl->members[g][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned g = s_group, m = s_member;
    if (m >= 32 || g >= (unsigned)ngroups) return;
    const int slot_floats = 32 * floats_per_wg;
    float* gring = ring + (size_t)g * 2 * slot_floats;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(gring, 0, 2 * slot_floats * 4, 0x00020000);
    unsigned retries = 0;
    float acc = 0.f;
    for (int s = 1; s <= steps; ++s) {
        float* slot = gring + (s & 1) * slot_floats + m * floats_per_wg;
        const float tag = (float)s + acc * 0.f;
        if (tid < floats_per_wg) {
            if (SMODE == 0) slot[tid] = tag;
            else __hip_atomic_store(&slot[tid], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(1);
        // every workgroup reads the whole slot: 512 threads x 16 B per pass; a fragment that is not yet step s is re-loaded
        const int base = (s & 1) * slot_floats * 4;
        for (int off = tid * 16; off < slot_floats * 4; off += 512 * 16) {
            for (;;) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs0, base + off, 0, LAUX);
                const f32x4 f = __builtin_bit_cast(f32x4, v);
                const bool ok = (f[0] == (float)s) & (f[1] == (float)s) & (f[2] == (float)s) & (f[3] == (float)s);
                if (__all(ok)) { acc += f[0]; break; }
                ++retries;
                if (retries > 2000000u) { acc = -1.f; break; }     // never hang the box: give up (shows as absurd retries)
                __builtin_amdgcn_s_sleep(1);
            }
        }
        float w = acc;
        for (int i = 0; i < work; ++i) w = __builtin_fmaf(w, 1.0000001f, 1e-9f);   // ~4-8 cycles each, dependent
        acc = w;
        __syncthreads();
    }
    if ((tid & 63) == 0) atomicAdd(&ctl->retries, retries);
    if (acc == 12345.678f) ring[0] = acc;
}

template <int SMODE, int LAUX>
void run(const char* name, Ctl* ctl, float* ring, int floats_per_wg, int by_block, int work, int delay, int ngroups = 6) {
    const int steps = 400;
    hipMemset(ctl, 0, sizeof(Ctl));
    hipMemset(ring, 0, (size_t)8 * 2 * 32 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    Ctl h;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(ctl, 0, sizeof(Ctl));
        hipMemset(ring, 0, (size_t)8 * 2 * 32 * 1024 * 4);
        hipEventRecord(e0);
        probe<SMODE, LAUX><<<256, 512>>>(ctl, ring, steps, floats_per_wg, by_block, ngroups, work, delay);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost);
    }
    printf("%-36s %5d B/wg work %4d delay %2d: %6.2f us/step  retries/wave/step %.2f  census=", name, floats_per_wg * 4, work,
           delay, best * 1e3 / steps, h.retries / (double)(ngroups * 32 * 8 * steps));
    for (int i = 0; i < 8; ++i) printf("%u ", h.census[i]);
    printf("\n");
    fflush(stdout);
}

int main(int argc, char** argv) {
    Ctl* ctl;
    float* ring;
    hipMalloc(&ctl, sizeof(Ctl));
    // ring memory flavour: default hipMalloc (coarse-grained); "fine" = hipDeviceMallocFinegrained; "uncached" = hipDeviceMallocUncached
    const char* flavour = argc > 1 ? argv[1] : "coarse";
    hipError_t err = hipSuccess;
    if (flavour[0] == 'f') err = hipExtMallocWithFlags((void**)&ring, (size_t)8 * 2 * 32 * 1024 * 4, hipDeviceMallocFinegrained);
    else if (flavour[0] == 'u') err = hipExtMallocWithFlags((void**)&ring, (size_t)8 * 2 * 32 * 1024 * 4, hipDeviceMallocUncached);
    else err = hipMalloc(&ring, (size_t)8 * 2 * 32 * 1024 * 4);
    printf("ring memory: %s (%s)\n", flavour, hipGetErrorString(err));
    if (err != hipSuccess) return 1;
    for (int fl : {96, 288})
        for (int work : {0})
            for (int delay : {0, 6, 12}) {
                run<0, 16>("same XCD, plain st, sc1 ld", ctl, ring, fl, 0, work, delay);
                run<0, 2>("same XCD, plain st, nt ld", ctl, ring, fl, 0, work, delay);
                run<0, 17>("same XCD, plain st, sc0 sc1 ld", ctl, ring, fl, 0, work, delay);
                run<1, 16>("same XCD, sc1 st, sc1 ld", ctl, ring, fl, 0, work, delay);
                run<1, 16>("consecutive blocks, sc1 st, sc1 ld", ctl, ring, fl, 1, work, delay);
                if (flavour[0] != 'c') {        // coherent memory flavours: plain accesses across XCDs too
                    run<0, 0>("consecutive blocks, plain st, plain ld", ctl, ring, fl, 1, work, delay);
                    run<0, 16>("consecutive blocks, plain st, sc1 ld", ctl, ring, fl, 1, work, delay);
                }
            }
    return 0;
}

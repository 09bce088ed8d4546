// Probe: per-step cost of an all-to-all hand-off among 32 workgroups when the group is (a) one XCD's 32 CUs with plain
// (L2-resident) stores, (b) one XCD with sc1 (write-through) stores, (c) spread over all XCDs with sc1 stores.
// 256 workgroups of 512 threads (1 per CU).  Census: every workgroup reads HW_REG_XCC_ID and takes a member slot.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Ctl {
    unsigned int members[8][32];   // per-XCD member counters (own line each)
    unsigned int total[32];
    unsigned int arrive[8][8][32];  // per-group arrival counters, 8 shards each on its own line
    unsigned int bad;              // stale reads seen
    unsigned int census[8];
};

__device__ __forceinline__ unsigned xcc_id() {
    return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xF;   // HW_REG_XCC_ID = 20, bits [3:0]
}

template <int MODE, int LAUX = 16, int LOCALSYNC = 0>   // LOCALSYNC: arrival add and poll without sc1 (XCD L2 only); MODE 0: plain stores, 1: sc1 stores, 2: nt stores; LAUX: load aux bits (16 sc1, 2 nt, 1 sc0)
__global__ __launch_bounds__(512) void probe(Ctl* ctl, float* ring, int steps, int floats_per_wg, int by_block, int ngroups) {
    __shared__ unsigned s_group, s_member;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        unsigned g, m;
        if (by_block) {
            g = blockIdx.x >> 5;          // 32 consecutive blocks = one group: spread over all 8 XCDs
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
    unsigned bad = 0;
    for (int s = 1; s <= steps; ++s) {
        float* slot = gring + (s & 1) * slot_floats + m * floats_per_wg;
        if (tid < floats_per_wg) {
            if (MODE == 0) slot[tid] = (float)s;
            else if (MODE == 2) __builtin_nontemporal_store((float)s, &slot[tid]);
            else __hip_atomic_store(&slot[tid], (float)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (LOCALSYNC) __hip_atomic_fetch_add(&ctl->arrive[g][m & 7][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(&ctl->arrive[g][m & 7][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (wave == 0) {
            const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(&ctl->arrive[g][0][0], 0, 8 * 32 * 4, 0x00020000);
            for (;;) {
                bool ok = true;
                if (lane < 8) {
                    unsigned v;
                    if (LOCALSYNC) v = __builtin_amdgcn_raw_buffer_load_b32(rsc, lane * 128, 0, LOCALSYNC == 1 ? 2 : 16);   // nt or sc1 poll
                    else v = __hip_atomic_load(&ctl->arrive[g][lane][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = v >= 4u * s;
                }
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        // every workgroup reads the whole slot: 512 threads x 16 B per pass
        const int base = (s & 1) * slot_floats * 4;
        for (int off = tid * 16; off < slot_floats * 4; off += 512 * 16) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs0, base + off, 0, LAUX);
            const f32x4 f = __builtin_bit_cast(f32x4, v);
            bad += (f[0] != (float)s) + (f[1] != (float)s) + (f[2] != (float)s) + (f[3] != (float)s);
        }
    }
    if (bad) atomicAdd(&ctl->bad, bad);
}

template <int MODE, int LAUX = 16, int LOCALSYNC = 0>
void run(const char* name, Ctl* ctl, float* ring, int floats_per_wg, int by_block, int ngroups = 6) {
    const int steps = 400;
    hipMemset(ctl, 0, sizeof(Ctl));
    hipMemset(ring, 0, (size_t)8 * 2 * 32 * floats_per_wg * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<MODE, LAUX, LOCALSYNC><<<256, 512>>>(ctl, ring, steps, floats_per_wg, by_block, ngroups);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    Ctl h; hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost);
    printf("%-44s %5d B/wg: %6.2f us/step  stale=%u  census=", name, floats_per_wg * 4, ms * 1e3 / steps, h.bad);
    for (int i = 0; i < 8; ++i) printf("%u ", h.census[i]);
    printf("\n");
}

int main() {
    Ctl* ctl; float* ring;
    hipMalloc(&ctl, sizeof(Ctl));
    hipMalloc(&ring, (size_t)8 * 2 * 32 * 1024 * 4);
    for (int fl : {100, 300}) {
        run<0, 2>("same XCD, plain st, nt ld, agent sync", ctl, ring, fl, 0);
        run<0, 2>("same XCD, plain st, nt ld, agent sync", ctl, ring, fl, 0);
        run<0, 2, 1>("same XCD, plain st, nt ld, L2 add + nt poll", ctl, ring, fl, 0);
        run<0, 2, 2>("same XCD, plain st, nt ld, L2 add + sc1 poll", ctl, ring, fl, 0);
        run<0, 16, 2>("same XCD, plain st, sc1 ld, L2 add + sc1 poll", ctl, ring, fl, 0);
        run<1, 16>("consecutive blocks, sc1 st, sc1 ld, agent", ctl, ring, fl, 1);
    }
    return 0;
}

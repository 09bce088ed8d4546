// Probe (round 4): would TWO independent recurrences per CU, software-pipelined, hide the hand-off round trips?
// The forward recurrence's exchange pattern in miniature: 240 workgroups of 512 threads (one per CU, forced by LDS), 6 groups of
// 40 members; per step a member stores its share of a 12.8 KB slot (write-through) and loads the WHOLE slot (sc1, 13 wave-loads
// of 1 KB), validates it (payload = step number: an old value means "not there yet", re-load), spends `work` cycles (dependent
// FMAs: the MFMA + gate phase), stores.  NREC = 1: what the kernel does today.  NREC = 2: every workgroup is a member of two
// groups (g and g + 3) with HALF the share and half the work in each, and runs them alternately -- one's loads are in flight
// while the other computes.  Same total work and the same stores per round; twice the bytes loaded.
//   hipcc --offload-arch=gfx950 -O2 tools/interleave_probe.hip -o tools/interleave_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int SLOT_FLOATS = 3200, NGROUP = 6, NMEM = 40;

__device__ __forceinline__ float spin_work(float w, int n) {
    for (int i = 0; i < n; ++i) w = __builtin_fmaf(w, 1.0000001f, 1e-9f);
    return w;
}

template <int NREC>
__global__ __launch_bounds__(512) void probe(float* ring, unsigned* retries_out, int steps, int work, int delay) {
    extern __shared__ float pad[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g0 = blockIdx.x / NMEM, m = blockIdx.x % NMEM;
    unsigned retries = 0;
    float acc = 0.f;
    // recurrence r of this workgroup: group (g0 + 3 r) % 6; member index m (+ 40 for the second role of a group)
    auto slot_of = [&](int r, int s) { return ring + ((size_t)((g0 + 3 * r) % NGROUP) * 2 + (s & 1)) * SLOT_FLOATS; };
    const int share = SLOT_FLOATS / (NMEM * NREC);                  // floats this workgroup contributes to a slot
    auto store = [&](int r, int s) {
        float* p = slot_of(r, s) + (m * NREC + r) * share;
        if (tid < share) __hip_atomic_store(&p[tid], (float)s + acc * 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    f32x4 frag[2][2];
    auto issue = [&](int r, int s) {
        for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(1);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slot_of(r, s), 0, SLOT_FLOATS * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kb = wave + 8 * i;                           // 13 KB: waves 0..4 take two wave-loads, 5..7 one
            frag[r][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, kb < 13 ? (kb * 256 + lane * 4) * 4 : 0x7FFFFFF0, 0, 16));
        }
    };
    auto validate = [&](int r, int s) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slot_of(r, s), 0, SLOT_FLOATS * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kb = wave + 8 * i;
            if (kb >= 13) continue;
            for (;;) {
                const f32x4 f = frag[r][i];
                const bool ok = ((kb * 256 + lane * 4) >= SLOT_FLOATS) | ((f[0] == (float)s) & (f[3] == (float)s));
                if (__all(ok)) break;
                if (++retries > 4000000u) break;                   // never hang the box
                __builtin_amdgcn_s_sleep(1);
                frag[r][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (kb * 256 + lane * 4) * 4, 0, 16));
            }
            acc += frag[r][i][0] * 0.f;
        }
    };
    for (int r = 0; r < NREC; ++r) store(r, 0);
    __syncthreads();
    issue(0, 0);
    for (int s = 0; s < steps; ++s) {
        if (NREC == 2) issue(1, s);
        validate(0, s);
        acc = spin_work(acc, work / NREC);
        __syncthreads();
        store(0, s + 1);
        __syncthreads();
        issue(0, s + 1);
        if (NREC == 2) {
            validate(1, s);
            acc = spin_work(acc, work / NREC);
            __syncthreads();
            store(1, s + 1);
            __syncthreads();
        }
    }
    if (lane == 0) atomicAdd(retries_out, retries);
    if (acc == 12345.678f) ring[0] = acc + pad[0];
}

int main() {
    float* ring; unsigned* retr;
    hipMalloc(&ring, (size_t)NGROUP * 2 * SLOT_FLOATS * 4);
    hipMalloc(&retr, 4);
    const int steps = 400;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int work : {0, 150, 300})
        for (int delay : {0, 4, 8, 12})
            for (int nrec : {1, 2}) {
                float best = 1e9f; unsigned h = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    hipMemset(ring, 0xFF, (size_t)NGROUP * 2 * SLOT_FLOATS * 4);
                    hipMemset(retr, 0, 4);
                    hipEventRecord(e0);
                    if (nrec == 1) probe<1><<<240, 512, 100 * 1024>>>(ring, retr, steps, work, delay);
                    else probe<2><<<240, 512, 100 * 1024>>>(ring, retr, steps, work, delay);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                    hipMemcpy(&h, retr, 4, hipMemcpyDeviceToHost);
                }
                printf("recurrences per CU %d  work %4d fma (~%.2f us)  first-attempt delay %2d sleeps: %6.2f us per round  re-loads/wave/round %.2f\n",
                       nrec, work, work * 4.0 / 2400.0 * (nrec == 1 ? 1 : 1), delay, best * 1e3 / steps, h / (240.0 * 8 * steps));
                fflush(stdout);
            }
    return 0;
}

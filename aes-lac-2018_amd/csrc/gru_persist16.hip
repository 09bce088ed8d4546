// Persistent BiGRU recurrence, the 16x16 MFMA forms (round 1: whole batch per workgroup; round 2: two batch parts; round 3: two
// parts on the bf16 matrix pipe with split operands) -- the forms B >= 13 (forward) / B >= 17 (backward) and every width the
// 4x4x1 forms of gru_persist.hip do not cover run.  Split out of gru_persist.hip in round 6; the protocol description, the
// exchange workspace and the C entry points are there.
#include "gru_persist_common.h"
#include "split_bf16.h"

namespace {

template <int NBT, int KBW>
__global__ __launch_bounds__(NWP * 64) void gru_fwd_persistent_kernel(float* __restrict__ G, float* __restrict__ ghn,
                                                                      float* __restrict__ hout,
                                                                      const float* __restrict__ w_hh,
                                                                      SyncWs* __restrict__ sync, float* __restrict__ ring,
                                                                      int T, int B, int H, int dbg) {
    __shared__ float red[NWP][2][NBT][16][17];
    __shared__ int abort_flag;

    const int tid = threadIdx.x, lane = tid & 63;
    // readfirstlane makes everything derived from the wave id provably wave-uniform: uniform branches and SGPR
    // buffer descriptors instead of per-load waterfall loops (cdna_hip_programming.md T20)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    unsigned int* shards = &sync->arrive[dir][0][0][0];
    unsigned int* ctr = shards + (blockIdx.x % NSHARD) * 32;
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        const int t = dir == 0 ? s : T - 1 - s;
        float gi_r = 0.f, gi_z = 0.f, gi_n = 0.f, sv_r = 0.f, sv_z = 0.f, sv_n = 0.f, sv_g = 0.f, sv_h = 0.f;
        size_t gbase = 0;
        if (gate_ok) {                                  // independent of h: issue before the wait
            gbase = (((size_t)t * B + gb) * 2 + dir) * 3 * H + gj;
            gi_r = G[gbase];
            gi_z = G[gbase + H];
            gi_n = G[gbase + 2 * H];
        }
        if (s > 0) {
            if (!DS2_DBG(dbg, 1) && wave == 0 && !wait_arrivals(shards, s, nslice, lane, &sync->error) && lane == 0)
                abort_flag = 1;
            __syncthreads();
            if (abort_flag) return;
            // h_{t-1} comes from the exchange ring, laid out [batch tile][k block][k quad 4][16 batch rows][4 k]: the
            // MFMA B fragment of lane l is bytes 16 l .. 16 l + 15 of ONE contiguous kilobyte, so a wave-load is eight
            // whole 128-B lines read in lane order (the (b, k) layout of hout gives 10-16 scattered 64-B pieces per
            // load and ran the CU's inbound path at ~27 GB/s; a [16 batch][16 k] block is contiguous per wave but
            // each 16-lane quarter still gathers four 64-B pieces and measured ~1 us/step slower)
            const int slot_floats = NBT * nkb * 256;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                ring + ((size_t)dir * 2 + ((s - 1) & 1)) * slot_floats, 0, slot_floats * 4, 0x00020000);
            // batch tiles are software-pipelined: the h fragments of tile bt+1 are in flight while the MFMAs of
            // tile bt issue (fully unrolled so the two fragment buffers stay in registers)
            f32x4 bf[2][KBW];
            auto fetch = [&](int bt, f32x4 (&dst)[KBW]) {
#pragma unroll
                for (int i = 0; i < KBW; ++i) {
                    const int kb = wave + NWP * i;                 // wave-uniform
                    // rows of padding batch entries hold whatever an earlier launch left: their products land in
                    // output columns >= B, which no gate thread reads
                    // k blocks past the end: an offset beyond the descriptor's range reads as zero, with no branch
                    dst[i] = LOAD_HANDOFF(rsrc, (kb < nkb) ? ((bt * nkb + kb) * 256 + lane * 4) * 4 : OOB_OFFSET);
                }
            };
            if (!DS2_DBG(dbg, 2)) {
                fetch(0, bf[0]);
#pragma unroll
                for (int bt = 0; bt < NBT; ++bt) {
                    if (bt + 1 < NBT) fetch(bt + 1, bf[(bt + 1) & 1]);
                    // all loads out before the MFMAs (the scheduler would otherwise sink them in between, 2 in flight)
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
                    for (int i = 0; i < KBW; ++i) {
                        // the step is MFMA-issue bound: a wave whose last k block lies past H (H = 800: 50 blocks
                        // over 8 waves, six of them own 6 not 7) skips its 8 all-zero MFMAs (wave-uniform branch)
                        if (i == KBW - 1 && wave + NWP * i >= nkb) break;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[0][i][e], bf[bt & 1][i][e], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[1][i][e], bf[bt & 1][i][e], acc1, 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        red[wave][0][bt][4 * q + r][m] = acc0[r];
                        red[wave][1][bt][4 * q + r][m] = acc1[r];
                    }
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
            const float r = fast_sigmoid(gi_r + gh_r);
            const float z = fast_sigmoid(gi_z + gh_z);
            const float n = fast_tanh(gi_n + r * gh_n);
            const float h = (1.f - z) * n + z * hp;
            hp = h;
            // handed to every other workgroup through the ring; hout keeps the plain copy for later launches
            store_sc1(&ring[(((size_t)dir * 2 + (s & 1)) * NBT * nkb + (size_t)gbt * nkb + (gj >> 4)) * 256 +
                            ((gj & 15) >> 2) * 64 + nn * 4 + (gj & 3)], h);
            sv_h = h;
            sv_r = r;
            sv_z = z;
            sv_n = n;
            sv_g = gh_n;
        }
        if (!DS2_DBG(dbg, 4)) wait_vmcnt0();   // every storing wave drains its hand-off store
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gate_ok) {   // saved activations are only read by later launches: keep them off the hand-off's critical path
            hout[(((size_t)dir * T + t) * B + gb) * H + gj] = sv_h;
            G[gbase] = sv_r;
            G[gbase + H] = sv_z;
            G[gbase + 2 * H] = sv_n;
            ghn[(((size_t)t * B + gb) * 2 + dir) * H + gj] = sv_g;
        }
    }
    if (tid == 0) leave_kernel(sync);
}

// Forward recurrence, 16x16x4 form with TWO batch parts (blockIdx.z): a workgroup owns 16 units and half the batch, so
// its gate rows fill three whole tiles (r16, z16, n16; the 8-unit form above pads its second tile by half) and it loads
// half the hidden state per step.  For B >= 24: per wave and batch tile 3 x KBW x 4 MFMAs instead of 2 x 2 x KBW x 4 for
// the two tiles a whole batch of 32 needs.  NBT = batch tiles of 16 per part (1 or 2).
template <int NBT, int KBW>
__global__ __launch_bounds__(NWP * 64) void gru_fwd_persistent_p2_kernel(float* __restrict__ G, float* __restrict__ ghn,
                                                                         float* __restrict__ hout,
                                                                         const float* __restrict__ w_hh,
                                                                         SyncWs* __restrict__ sync,
                                                                         float* __restrict__ ring, int T, int B, int H,
                                                                         int dbg) {
    __shared__ float red[NWP][3][NBT][16][17];
    __shared__ int abort_flag;
    constexpr int UNITS = 16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y, part = blockIdx.z, nslice = gridDim.x;
    const int j0 = blockIdx.x * UNITS;
    const int m = lane & 15, q = lane >> 4;
    const int nkb = H >> 4;
    const int bper = (B + 1) / 2, b0 = part * bper, nb = min(bper, B - b0);
    if (nb <= 0) {
        if (tid == 0) leave_kernel(sync);
        return;
    }
    const int slot_floats = NBT * nkb * 256;
    float* my_ring = ring + (size_t)(dir * 2 + part) * 2 * slot_floats;
    if (tid == 0) abort_flag = 0;

    f32x4 wreg[3][KBW];                                 // tile g = gate g of units j0 .. j0 + 15 (row m)
    {
        const bool unit_ok = (j0 + m) < H;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float* row = w_hh + ((size_t)dir * 3 * H + (size_t)g * H + (unit_ok ? j0 + m : 0)) * H;
#pragma unroll
            for (int i = 0; i < KBW; ++i) {
                const int kb = wave + NWP * i;
                wreg[g][i] = (kb < nkb && unit_ok) ? *reinterpret_cast<const f32x4*>(row + kb * 16 + q * 4)
                                                   : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    // gate role: unit jj, local batch row 16 gbt + nn  (512 threads = 16 units x 32 rows)
    const int jj = tid & 15, nn = (tid >> 4) & 15, gbt = tid >> 8;
    const int lb = gbt * 16 + nn, gb = b0 + lb, gj = j0 + jj;
    const bool gate_ok = (gbt < NBT) && (lb < nb) && (gj < H);
    float hp = 0.f;
    unsigned int* shards = &sync->arrive[dir][part][0][0];
    unsigned int* ctr = shards + (blockIdx.x % NSHARD) * 32;
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        const int t = dir == 0 ? s : T - 1 - s;
        float gi_r = 0.f, gi_z = 0.f, gi_n = 0.f, sv_r = 0.f, sv_z = 0.f, sv_n = 0.f, sv_g = 0.f, sv_h = 0.f;
        size_t gbase = 0;
        if (gate_ok) {                                  // independent of h: issue before the wait
            gbase = (((size_t)t * B + gb) * 2 + dir) * 3 * H + gj;
            gi_r = G[gbase];
            gi_z = G[gbase + H];
            gi_n = G[gbase + 2 * H];
        }
        if (s > 0) {
            if (!DS2_DBG(dbg, 1) && wave == 0 && !wait_arrivals(shards, s, nslice, lane, &sync->error) && lane == 0)
                abort_flag = 1;
            __syncthreads();
            if (abort_flag) return;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                my_ring + (size_t)((s - 1) & 1) * slot_floats, 0, slot_floats * 4, 0x00020000);
            f32x4 bf[2][KBW];
            auto fetch = [&](int bt, f32x4 (&dst)[KBW]) {
#pragma unroll
                for (int i = 0; i < KBW; ++i) {
                    const int kb = wave + NWP * i;                 // wave-uniform
                    dst[i] = LOAD_HANDOFF(rsrc, (kb < nkb) ? ((bt * nkb + kb) * 256 + lane * 4) * 4 : OOB_OFFSET);
                }
            };
            if (!DS2_DBG(dbg, 2)) {
                fetch(0, bf[0]);
#pragma unroll
                for (int bt = 0; bt < NBT; ++bt) {
                    if (bt + 1 < NBT) fetch(bt + 1, bf[(bt + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);             // all loads out before the MFMAs
                    f32x4 acc[3];
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < KBW; ++i) {
                        if (i == KBW - 1 && wave + NWP * i >= nkb) break;   // all-zero padded k block
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int g = 0; g < 3; ++g)
                                acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[g][i][e], bf[bt & 1][i][e], acc[g], 0, 0, 0);
                    }
#pragma unroll
                    for (int g = 0; g < 3; ++g)
#pragma unroll
                        for (int r = 0; r < 4; ++r) red[wave][g][bt][4 * q + r][m] = acc[g][r];
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
                    gh_z += red[w][1][gbt][jj][nn];
                    gh_n += red[w][2][gbt][jj][nn];
                }
            }
            const float r = fast_sigmoid(gi_r + gh_r);
            const float z = fast_sigmoid(gi_z + gh_z);
            const float n = fast_tanh(gi_n + r * gh_n);
            const float h = (1.f - z) * n + z * hp;
            hp = h;
            sv_h = h;
            sv_r = r;
            sv_z = z;
            sv_n = n;
            sv_g = gh_n;
        }
        {
            // exchange ring of this (direction, part): [batch tile][k block][k quad 4][16 batch rows][4 k].  Four neighbouring
            // gate threads (units 4u .. 4u+3 of one batch row: adjacent lanes, 16 contiguous bytes of the ring) hand their
            // values to the first of them, which issues ONE 16-byte write-through store: a quarter of the fabric writes
            // (a 4-byte sc1 store costs about six times a 16-byte one per byte).  H % 16 == 0: a quad never straddles H.
            f32x4 hq;
            hq[0] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sv_h), 0x00, 0xF, 0xF, true));
            hq[1] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sv_h), 0x55, 0xF, 0xF, true));
            hq[2] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sv_h), 0xAA, 0xF, 0xF, true));
            hq[3] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sv_h), 0xFF, 0xF, 0xF, true));
            if (gate_ok && (jj & 3) == 0) {
                const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(my_ring, 0, 2 * slot_floats * 4, 0x00020000);
                store_sc1_b128(rs_w, ((s & 1) * slot_floats + (gbt * nkb + (gj >> 4)) * 256 + ((gj & 15) >> 2) * 64 + nn * 4) * 4,
                               __builtin_bit_cast(u32x4, hq));
            }
        }
        if (!DS2_DBG(dbg, 4)) wait_vmcnt0();
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gate_ok) {
            hout[(((size_t)dir * T + t) * B + gb) * H + gj] = sv_h;
            G[gbase] = sv_r;
            G[gbase + H] = sv_z;
            G[gbase + 2 * H] = sv_n;
            ghn[(((size_t)t * B + gb) * 2 + dir) * H + gj] = sv_g;
        }
    }
    if (tid == 0) leave_kernel(sync);
}

// ----------------------------------------------------------------------------------------------------------
// The two-part 16x16 forms on the bf16 matrix pipe (B >= 17, H % 32 == 0).  From B = 17 half of a recurrence step is
// matrix-pipe time (16 units x 16 batch rows x 3 H k per workgroup on v_mfma_f32_16x16x4_f32: 84 / 76 instructions of 32
// cycles per wave and batch tile).  Here every fp32 operand is split without error into three bf16 terms (split_bf16.h) and
// the products run on v_mfma_f32_16x16x32_bf16 -- six exact partial products per 32 k, 16 cycles each: 0.43 x the pipe time
// for the same fp32 result (the accumulator is fp32 as before).  The weights are split once per launch into registers
// (three planes of 8 bf16 per lane and 32-k block: 1.5 x the registers of the fp32 copy); the NEW state is split by the
// gate thread that produces it and written to the exchange ring as three bf16 planes, eight units per 16-byte write-through
// store (two neighbours' values are packed by one split, the four dwords of an octet gathered with DPP) -- 1.5 x the
// hand-off bytes, no conversion on the consumers' side.
// Ring layout of a (direction, part): [slot 2][batch tile][k block of 32][plane 3][k octet 4][16 batch rows][8 k] bf16: a
// wave-load of one plane of one block is 1 KB contiguous, lane l = (batch row l & 15, octet l >> 4).
// ----------------------------------------------------------------------------------------------------------
typedef __bf16 gbf16x8 __attribute__((ext_vector_type(8)));

// the octet gather of one plane: lanes 8 o + {0, 2, 4, 6} hold the packed pairs; lane 8 o receives all four
__device__ __forceinline__ u32x4 gather_octet(unsigned int pair) {
    const int p = (int)pair;
    const int x1 = __builtin_amdgcn_update_dpp(0, p, 0xAA, 0xF, 0xF, true);      // quad lane 2's pair
    const int x2 = __builtin_amdgcn_update_dpp(0, p, 0x104, 0xF, 0xF, true);     // row_shl:4 -> lane + 4's pair
    const int x3 = __builtin_amdgcn_update_dpp(0, x1, 0x104, 0xF, 0xF, true);    // lane + 6's pair
    u32x4 v;
    v[0] = pair;
    v[1] = (unsigned int)x1;
    v[2] = (unsigned int)x2;
    v[3] = (unsigned int)x3;
    return v;
}
// split this thread's value together with its odd neighbour's (lanes 2 i, 2 i + 1 hold units 2 i, 2 i + 1 of one batch row)
// and gather the octet: planes[q] is valid in lanes with (unit & 7) == 0
__device__ __forceinline__ void split_gather_octet(float v, u32x4 (&planes)[3]) {
    const float nb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // lane ^ 1
    unsigned int p1, p2, p3;
    split3(v, nb, p1, p2, p3);
    planes[0] = gather_octet(p1);
    planes[1] = gather_octet(p2);
    planes[2] = gather_octet(p3);
}

template <int NBT, int KBW>
__global__ __launch_bounds__(NWP * 64) void gru_fwd_persistent_p2b_kernel(float* __restrict__ G, float* __restrict__ ghn,
                                                                          float* __restrict__ hout,
                                                                          const float* __restrict__ w_hh,
                                                                          SyncWs* __restrict__ sync,
                                                                          float* __restrict__ ring, int T, int B, int H,
                                                                          int dbg) {
    __shared__ float red[NWP][3][NBT][16][17];
    __shared__ int abort_flag;
    constexpr int UNITS = 16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y, part = blockIdx.z, nslice = gridDim.x;
    const int j0 = blockIdx.x * UNITS;
    const int m = lane & 15, q = lane >> 4;
    const int nkb = H >> 5;                             // k blocks of 32
    const int bper = (B + 1) / 2, b0 = part * bper, nb = min(bper, B - b0);
    if (nb <= 0) {
        if (tid == 0) leave_kernel(sync);
        return;
    }
    const int slot_bytes = NBT * nkb * 3 * 1024;
    char* my_ring = reinterpret_cast<char*>(ring) + (size_t)(dir * 2 + part) * 2 * slot_bytes;
    if (tid == 0) abort_flag = 0;

    gbf16x8 wq[3][KBW][3];                              // [gate][k block][plane]: row m = unit j0 + m, k = 32 kb + 8 q ..
    {
        const bool unit_ok = (j0 + m) < H;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float* row = w_hh + ((size_t)dir * 3 * H + (size_t)g * H + (unit_ok ? j0 + m : 0)) * H;
#pragma unroll
            for (int i = 0; i < KBW; ++i) {
                const int kb = wave + NWP * i;
                f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
                if (kb < nkb && unit_ok) {
                    lo = *reinterpret_cast<const f32x4*>(row + kb * 32 + q * 8);
                    hi = *reinterpret_cast<const f32x4*>(row + kb * 32 + q * 8 + 4);
                }
                unsigned int pl[3][4];
                split3(lo[0], lo[1], pl[0][0], pl[1][0], pl[2][0]);
                split3(lo[2], lo[3], pl[0][1], pl[1][1], pl[2][1]);
                split3(hi[0], hi[1], pl[0][2], pl[1][2], pl[2][2]);
                split3(hi[2], hi[3], pl[0][3], pl[1][3], pl[2][3]);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const u32x4 v = {pl[c][0], pl[c][1], pl[c][2], pl[c][3]};
                    wq[g][i][c] = __builtin_bit_cast(gbf16x8, v);
                }
            }
        }
    }
    // gate role: unit jj, local batch row 16 gbt + nn  (512 threads = 16 units x 32 rows)
    const int jj = tid & 15, nn = (tid >> 4) & 15, gbt = tid >> 8;
    const int lb = gbt * 16 + nn, gb = b0 + lb, gj = j0 + jj;
    const bool gate_ok = (gbt < NBT) && (lb < nb) && (gj < H);
    float hp = 0.f;
    unsigned int* shards = &sync->arrive[dir][part][0][0];
    unsigned int* ctr = shards + (blockIdx.x % NSHARD) * 32;
    __syncthreads();

    // Round 4 (as in the 4x4x1 kernels): the gate pre-activations of step s + 1 are loaded at the END of step s, behind the
    // saved-activation stores, instead of at the top of step s + 1 in front of its hand-off loads (a wave's vector-memory
    // operations complete in issue order)
    float gi_r = 0.f, gi_z = 0.f, gi_n = 0.f;
    auto load_gi = [&](int t) {
        if (gate_ok) {
            const size_t gb3 = (((size_t)t * B + gb) * 2 + dir) * 3 * H + gj;
            gi_r = G[gb3];
            gi_z = G[gb3 + H];
            gi_n = G[gb3 + 2 * H];
        }
    };
    load_gi(dir == 0 ? 0 : T - 1);
    for (int s = 0; s < T; ++s) {
        const int t = dir == 0 ? s : T - 1 - s;
        float sv_r = 0.f, sv_z = 0.f, sv_n = 0.f, sv_g = 0.f, sv_h = 0.f;
        const size_t gbase = (((size_t)t * B + gb) * 2 + dir) * 3 * H + gj;
        if (s > 0) {
            if (!DS2_DBG(dbg, 1) && wave == 0 && !wait_arrivals(shards, s, nslice, lane, &sync->error) && lane == 0)
                abort_flag = 1;
            __syncthreads();
            if (abort_flag) return;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                my_ring + (size_t)((s - 1) & 1) * slot_bytes, 0, slot_bytes, 0x00020000);
            if (!DS2_DBG(dbg, 2)) {
#pragma unroll
                for (int bt = 0; bt < NBT; ++bt) {
                    gbf16x8 bf[KBW][3];
#pragma unroll
                    for (int i = 0; i < KBW; ++i) {
                        const int kb = wave + NWP * i;                 // wave-uniform
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            bf[i][c] = __builtin_bit_cast(
                                gbf16x8, LOAD_HANDOFF(rsrc, (kb < nkb) ? ((bt * nkb + kb) * 3 + c) * 1024 + lane * 16 : OOB_OFFSET));
                    }
                    __builtin_amdgcn_sched_barrier(0);                 // all loads out before the MFMAs
                    f32x4 acc[3];
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < KBW; ++i) {
                        if (i == KBW - 1 && wave + NWP * i >= nkb) break;   // all-zero padded k block
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            f32x4 a = acc[g];
                            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[g][i][1], bf[i][1], a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[g][i][0], bf[i][2], a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[g][i][2], bf[i][0], a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[g][i][0], bf[i][1], a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[g][i][1], bf[i][0], a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[g][i][0], bf[i][0], a, 0, 0, 0);
                            acc[g] = a;
                        }
                    }
#pragma unroll
                    for (int g = 0; g < 3; ++g)
#pragma unroll
                        for (int r = 0; r < 4; ++r) red[wave][g][bt][4 * q + r][m] = acc[g][r];
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
                    gh_z += red[w][1][gbt][jj][nn];
                    gh_n += red[w][2][gbt][jj][nn];
                }
            }
            const float r = fast_sigmoid(gi_r + gh_r);
            const float z = fast_sigmoid(gi_z + gh_z);
            const float n = fast_tanh(gi_n + r * gh_n);
            const float h = (1.f - z) * n + z * hp;
            hp = h;
            sv_h = h;
            sv_r = r;
            sv_z = z;
            sv_n = n;
            sv_g = gh_n;
        }
        {
            u32x4 planes[3];
            split_gather_octet(sv_h, planes);                          // (every lane: DPP needs the whole row active)
            if (gate_ok && (jj & 7) == 0) {
                const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(my_ring, 0, 2 * slot_bytes, 0x00020000);
                const int base = (s & 1) * slot_bytes + (gbt * nkb + (gj >> 5)) * 3 * 1024 + ((gj & 31) >> 3) * 256 + nn * 16;
#pragma unroll
                for (int c = 0; c < 3; ++c) store_sc1_b128(rs_w, base + c * 1024, planes[c]);
            }
        }
        if (!DS2_DBG(dbg, 4)) wait_vmcnt0();
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gate_ok) {
            hout[(((size_t)dir * T + t) * B + gb) * H + gj] = sv_h;
            G[gbase] = sv_r;
            G[gbase + H] = sv_z;
            G[gbase + 2 * H] = sv_n;
            ghn[(((size_t)t * B + gb) * 2 + dir) * H + gj] = sv_g;
        }
        if (s + 1 < T) load_gi(dir == 0 ? s + 1 : T - 2 - s);
    }
    if (tid == 0) leave_kernel(sync);
}

template <int NBT, int KBW>
__global__ __launch_bounds__(NWP * 64) void gru_bwd_persistent_kernel(float* __restrict__ G, float* __restrict__ ghn,
                                                                      const float* __restrict__ hout,
                                                                      const float* __restrict__ d_out,
                                                                      const float* __restrict__ w_hh_t,
                                                                      SyncWs* __restrict__ sync, float* __restrict__ ring,
                                                                      int T, int B, int H, int dbg) {
    __shared__ float red[NWP][NBT][16][17];
    __shared__ int abort_flag;

    const int tid = threadIdx.x, lane = tid & 63;
    // readfirstlane makes everything derived from the wave id provably wave-uniform: uniform branches and SGPR
    // buffer descriptors instead of per-load waterfall loops (cdna_hip_programming.md T20)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    unsigned int* shards = &sync->arrive[dir][0][0][0];
    unsigned int* ctr = shards + (blockIdx.x % NSHARD) * 32;
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        const int t = dir == 0 ? T - 1 - s : s;
        const bool has_prev = dir == 0 ? (t > 0) : (t < T - 1);
        // saved activations of step t (written by the forward pass, an earlier launch): plain loads
        float dh = 0.f, r = 0.f, z = 0.f, n = 0.f, gn = 0.f, hpv = 0.f, sv_r = 0.f, sv_z = 0.f, sv_n = 0.f, sv_g = 0.f;
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
            if (!DS2_DBG(dbg, 1) && wave == 0 && !wait_arrivals(shards, s, nslice, lane, &sync->error) && lane == 0)
                abort_flag = 1;
            __syncthreads();
            if (abort_flag) return;
            // dGH of step tnext = [dr_pre | dz_pre | d(gh_n)] comes from the exchange ring, laid out
            // [batch tile][k block][16 batch rows][16 k]: one wave-load = one contiguous kilobyte
            const int slot_floats = NBT * nkb * 256;
            const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
                ring + ((size_t)dir * 2 + ((s - 1) & 1)) * slot_floats, 0, slot_floats * 4, 0x00020000);
            // stages = (batch tile, k chunk); the fragments of stage st+1 are fetched while stage st's MFMAs issue
            constexpr int CH = (NBT > 1 && KBW > 10) ? 7 : KBW;
            constexpr int NCH = (KBW + CH - 1) / CH;
            constexpr int NST = NBT * NCH;
            f32x4 bf[2][CH];
            auto fetch = [&](int st, f32x4 (&dst)[CH]) {
                const int bt = st / NCH, i0 = (st % NCH) * CH;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int i = i0 + c;
                    const int kb = wave + NWP * i;                 // wave-uniform
                    if (i < KBW)                                   // compile-time
                        dst[c] = LOAD_HANDOFF(rs_x, (kb < nkb) ? ((bt * nkb + kb) * 256 + lane * 4) * 4 : OOB_OFFSET);
                    else
                        dst[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            };
            if (!DS2_DBG(dbg, 2)) {
                fetch(0, bf[0]);
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    if (st + 1 < NST) fetch(st + 1, bf[(st + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);   // loads out before the MFMAs
                    const int bt = st / NCH, i0 = (st % NCH) * CH;
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        if (i0 + c < KBW) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[i0 + c][e], bf[st & 1][c][e], acc, 0, 0, 0);
                        }
                    if ((st % NCH) == NCH - 1) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) red[wave][bt][4 * q + rr][m] = acc[rr];
                        acc = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
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
            {   // hand-off copies into the ring (write-through); k index of unit j in gate g is g*H + j
                float* slot = ring + (((size_t)dir * 2 + (s & 1)) * NBT + gbt) * (size_t)nkb * 256 + nn * 4;
                const int k0 = gj, k1 = H + gj, k2 = 2 * H + gj;
                store_sc1(&slot[(size_t)(k0 >> 4) * 256 + ((k0 & 15) >> 2) * 64 + (k0 & 3)], dr_pre);
                store_sc1(&slot[(size_t)(k1 >> 4) * 256 + ((k1 & 15) >> 2) * 64 + (k1 & 3)], dz_pre);
                store_sc1(&slot[(size_t)(k2 >> 4) * 256 + ((k2 & 15) >> 2) * 64 + (k2 & 3)], dn_pre * r);
            }
            sv_r = dr_pre;
            sv_z = dz_pre;
            sv_n = dn_pre;
            sv_g = dn_pre * r;
        }
        if (!DS2_DBG(dbg, 4)) wait_vmcnt0();
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gate_ok) {   // d(gi), d(gh_n) for the GEMMs that follow this launch: plain stores, off the critical path
            G[gbase] = sv_r;
            G[gbase + H] = sv_z;
            G[gbase + 2 * H] = sv_n;
            ghn[row * H + gj] = sv_g;
        }
    }
    if (tid == 0) leave_kernel(sync);
}

// Backward recurrence, 16x16x4 form with TWO batch parts (blockIdx.z), the twin of gru_fwd_persistent_p2_kernel: a
// workgroup owns 16 units -- 16 rows of w_hh_t, a whole MFMA tile (the 8-unit form above pads half of every tile) -- and
// half the batch, so per step it pulls HALF of d(gh) (150 KB at B = 32 instead of 300) and issues NBT x 150 / 8 x 4 MFMAs
// per wave.  For B >= 17: the 4x4x1 forms' cost grows with every batch quad (B = 32: 7.0 us per step, B = 64: 12.3), this
// form's with every tile of 16 rows per part.  The hand-off loads of k chunk c + 1 are in flight under the MFMAs of chunk c.
template <int NBT, int KBW>
__global__ __launch_bounds__(NWP * 64) void gru_bwd_persistent_p2_kernel(float* __restrict__ G, float* __restrict__ ghn,
                                                                         const float* __restrict__ hout,
                                                                         const float* __restrict__ d_out,
                                                                         const float* __restrict__ w_hh_t,
                                                                         SyncWs* __restrict__ sync, float* __restrict__ ring,
                                                                         int T, int B, int H, int dbg) {
    __shared__ float red[NWP][NBT][16][17];
    __shared__ int abort_flag;
    constexpr int UNITS = 16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y, part = blockIdx.z, nslice = gridDim.x;
    const int j0 = blockIdx.x * UNITS;
    const int m = lane & 15, q = lane >> 4;
    const int K = 3 * H, nkb = K >> 4;
    const int bper = (B + 1) / 2, b0 = part * bper, nb = min(bper, B - b0);
    if (nb <= 0) {
        if (tid == 0) leave_kernel(sync);
        return;
    }
    const int slot_floats = NBT * nkb * 256;
    float* my_ring = ring + (size_t)(dir * 2 + part) * 2 * slot_floats;
    if (tid == 0) abort_flag = 0;

    f32x4 wreg[KBW];                                   // row m: column (j0 + m) of W_hh = row of w_hh_t
    {
        const bool unit_ok = (j0 + m) < H;
        const float* row = w_hh_t + ((size_t)dir * H + (unit_ok ? j0 + m : 0)) * K;
#pragma unroll
        for (int i = 0; i < KBW; ++i) {
            const int kb = wave + NWP * i;
            wreg[i] = (unit_ok && kb < nkb) ? *reinterpret_cast<const f32x4*>(row + kb * 16 + q * 4)
                                            : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // gate role: unit jj, local batch row 16 gbt + nn  (512 threads = 16 units x 32 rows)
    const int jj = tid & 15, nn = (tid >> 4) & 15, gbt = tid >> 8;
    const int lb = gbt * 16 + nn, gb = b0 + lb, gj = j0 + jj;
    const bool gate_ok = (gbt < NBT) && (lb < nb) && (gj < H);
    float dhz = 0.f;                                    // dh * z carried to the next (earlier) step
    unsigned int* shards = &sync->arrive[dir][part][0][0];
    unsigned int* ctr = shards + (blockIdx.x % NSHARD) * 32;
    __syncthreads();

    // Round 4 (as in the 4x4x1 kernels): the saved activations of step s + 1 are loaded at the END of step s, behind the
    // saved-activation stores, not at the top of step s + 1 in front of its hand-off loads
    float dh = 0.f, r = 0.f, z = 0.f, n = 0.f, gn = 0.f, hpv = 0.f;
    auto load_saved = [&](int t) {
        dh = r = z = n = gn = hpv = 0.f;
        if (gate_ok) {
            const bool has_prev = dir == 0 ? (t > 0) : (t < T - 1);
            const size_t row = ((size_t)t * B + gb) * 2 + dir, gbase = row * 3 * H + gj;
            dh = d_out[((size_t)t * B + gb) * H + gj];
            r = G[gbase];
            z = G[gbase + H];
            n = G[gbase + 2 * H];
            gn = ghn[row * H + gj];
            if (has_prev) hpv = hout[(((size_t)dir * T + (dir == 0 ? t - 1 : t + 1)) * B + gb) * H + gj];
        }
    };
    load_saved(dir == 0 ? T - 1 : 0);
    for (int s = 0; s < T; ++s) {
        const int t = dir == 0 ? T - 1 - s : s;
        float sv_r = 0.f, sv_z = 0.f, sv_n = 0.f, sv_g = 0.f;
        const size_t row = ((size_t)t * B + gb) * 2 + dir, gbase = row * 3 * H + gj;
        if (s > 0) {
            if (!DS2_DBG(dbg, 1) && wave == 0 && !wait_arrivals(shards, s, nslice, lane, &sync->error) && lane == 0)
                abort_flag = 1;
            __syncthreads();
            if (abort_flag) return;
            // d(gh) of the previous step of this (direction, part): [batch tile][k block][k quad 4][16 batch rows][4 k]
            const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
                my_ring + (size_t)((s - 1) & 1) * slot_floats, 0, slot_floats * 4, 0x00020000);
            constexpr int CH = KBW > 10 ? 7 : KBW;     // stages = (batch tile, k chunk)
            constexpr int NCH = (KBW + CH - 1) / CH;
            constexpr int NST = NBT * NCH;
            f32x4 bf[2][CH];
            auto fetch = [&](int st, f32x4 (&dst)[CH]) {
                const int bt = st / NCH, i0 = (st % NCH) * CH;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int i = i0 + c;
                    const int kb = wave + NWP * i;                 // wave-uniform
                    if (i < KBW)                                   // compile-time
                        dst[c] = LOAD_HANDOFF(rs_x, (kb < nkb) ? ((bt * nkb + kb) * 256 + lane * 4) * 4 : OOB_OFFSET);
                    else
                        dst[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            };
            if (!DS2_DBG(dbg, 2)) {
                fetch(0, bf[0]);
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    if (st + 1 < NST) fetch(st + 1, bf[(st + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);   // loads out before the MFMAs
                    const int bt = st / NCH, i0 = (st % NCH) * CH;
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        if (i0 + c < KBW) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[i0 + c][e], bf[st & 1][c][e], acc, 0, 0, 0);
                        }
                    if ((st % NCH) == NCH - 1) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) red[wave][bt][4 * q + rr][m] = acc[rr];
                        acc = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
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
            sv_r = dr_pre;
            sv_z = dz_pre;
            sv_n = dn_pre;
            sv_g = dn_pre * r;
        }
        {   // hand-off copies into the ring (write-through); k index of unit j in gate g is g*H + j.  Quads of gate threads
            // (units 4u .. 4u+3 of one batch row) gather their values into ONE 16-byte store per gate, as in the forward twin.
            float q3[3][4];
#define DS2_QUAD_BCAST(J)                                                                                              \
    q3[0][J] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sv_r), (J) * 0x55, 0xF, 0xF, true));        \
    q3[1][J] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sv_z), (J) * 0x55, 0xF, 0xF, true));        \
    q3[2][J] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sv_g), (J) * 0x55, 0xF, 0xF, true));
            DS2_QUAD_BCAST(0)
            DS2_QUAD_BCAST(1)
            DS2_QUAD_BCAST(2)
            DS2_QUAD_BCAST(3)
#undef DS2_QUAD_BCAST
            if (gate_ok && (jj & 3) == 0) {
                const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(my_ring, 0, 2 * slot_floats * 4, 0x00020000);
                const int sbase = (s & 1) * slot_floats + gbt * nkb * 256 + nn * 4;
#pragma unroll
                for (int g3 = 0; g3 < 3; ++g3) {
                    const int k = g3 * H + gj;
                    const f32x4 v = {q3[g3][0], q3[g3][1], q3[g3][2], q3[g3][3]};
                    store_sc1_b128(rs_w, (sbase + (k >> 4) * 256 + ((k & 15) >> 2) * 64) * 4, __builtin_bit_cast(u32x4, v));
                }
            }
        }
        if (!DS2_DBG(dbg, 4)) wait_vmcnt0();
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gate_ok) {   // d(gi), d(gh_n) for the GEMMs that follow this launch: plain stores, off the critical path
            G[gbase] = sv_r;
            G[gbase + H] = sv_z;
            G[gbase + 2 * H] = sv_n;
            ghn[row * H + gj] = sv_g;
        }
        if (s + 1 < T) load_saved(dir == 0 ? T - 2 - s : s + 1);
    }
    if (tid == 0) leave_kernel(sync);
}

// Backward twin of gru_fwd_persistent_p2b_kernel: K = 3 H (d(gh) of the previous step in the ring, as three bf16 planes),
// rows of w_hh_t split into registers once.  The hand-off fragments of a batch tile are fetched in chunks of CH k blocks
// (the weights take 12 registers per block, a chunk of fragments 12 per block too).
template <int NBT, int KBW>
__global__ __launch_bounds__(NWP * 64) void gru_bwd_persistent_p2b_kernel(float* __restrict__ G, float* __restrict__ ghn,
                                                                          const float* __restrict__ hout,
                                                                          const float* __restrict__ d_out,
                                                                          const float* __restrict__ w_hh_t,
                                                                          SyncWs* __restrict__ sync, float* __restrict__ ring,
                                                                          int T, int B, int H, int dbg) {
    __shared__ float red[NWP][NBT][16][17];
    __shared__ int abort_flag;
    constexpr int UNITS = 16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y, part = blockIdx.z, nslice = gridDim.x;
    const int j0 = blockIdx.x * UNITS;
    const int m = lane & 15, q = lane >> 4;
    const int K = 3 * H, nkb = K >> 5;                  // k blocks of 32
    const int bper = (B + 1) / 2, b0 = part * bper, nb = min(bper, B - b0);
    if (nb <= 0) {
        if (tid == 0) leave_kernel(sync);
        return;
    }
    const int slot_bytes = NBT * nkb * 3 * 1024;
    char* my_ring = reinterpret_cast<char*>(ring) + (size_t)(dir * 2 + part) * 2 * slot_bytes;
    if (tid == 0) abort_flag = 0;

    gbf16x8 wq[KBW][3];                                 // row m: column (j0 + m) of W_hh = row of w_hh_t, k = 32 kb + 8 q ..
    {
        const bool unit_ok = (j0 + m) < H;
        const float* row = w_hh_t + ((size_t)dir * H + (unit_ok ? j0 + m : 0)) * K;
#pragma unroll
        for (int i = 0; i < KBW; ++i) {
            const int kb = wave + NWP * i;
            f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
            if (kb < nkb && unit_ok) {
                lo = *reinterpret_cast<const f32x4*>(row + kb * 32 + q * 8);
                hi = *reinterpret_cast<const f32x4*>(row + kb * 32 + q * 8 + 4);
            }
            unsigned int pl[3][4];
            split3(lo[0], lo[1], pl[0][0], pl[1][0], pl[2][0]);
            split3(lo[2], lo[3], pl[0][1], pl[1][1], pl[2][1]);
            split3(hi[0], hi[1], pl[0][2], pl[1][2], pl[2][2]);
            split3(hi[2], hi[3], pl[0][3], pl[1][3], pl[2][3]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const u32x4 v = {pl[c][0], pl[c][1], pl[c][2], pl[c][3]};
                wq[i][c] = __builtin_bit_cast(gbf16x8, v);
            }
        }
    }
    const int jj = tid & 15, nn = (tid >> 4) & 15, gbt = tid >> 8;
    const int lb = gbt * 16 + nn, gb = b0 + lb, gj = j0 + jj;
    const bool gate_ok = (gbt < NBT) && (lb < nb) && (gj < H);
    float dhz = 0.f;                                    // dh * z carried to the next (earlier) step
    unsigned int* shards = &sync->arrive[dir][part][0][0];
    unsigned int* ctr = shards + (blockIdx.x % NSHARD) * 32;
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        const int t = dir == 0 ? T - 1 - s : s;
        const bool has_prev = dir == 0 ? (t > 0) : (t < T - 1);
        float dh = 0.f, r = 0.f, z = 0.f, n = 0.f, gn = 0.f, hpv = 0.f, sv_r = 0.f, sv_z = 0.f, sv_n = 0.f, sv_g = 0.f;
        size_t row = 0, gbase = 0;
        if (gate_ok) {                                  // saved activations of step t: plain loads, issued before the wait
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
            if (!DS2_DBG(dbg, 1) && wave == 0 && !wait_arrivals(shards, s, nslice, lane, &sync->error) && lane == 0)
                abort_flag = 1;
            __syncthreads();
            if (abort_flag) return;
            const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
                my_ring + (size_t)((s - 1) & 1) * slot_bytes, 0, slot_bytes, 0x00020000);
            constexpr int CH = KBW > 2 ? 2 : KBW;      // stages = (batch tile, k chunk), double-buffered
            constexpr int NCH = (KBW + CH - 1) / CH;
            constexpr int NST = NBT * NCH;
            if (!DS2_DBG(dbg, 2)) {
                gbf16x8 bf[2][CH][3];
                auto fetch = [&](int st, gbf16x8 (&dst)[CH][3]) {
                    const int bt = st / NCH, i0 = (st % NCH) * CH;
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const int i = i0 + c;
                        const int kb = wave + NWP * i;                 // wave-uniform
#pragma unroll
                        for (int pq = 0; pq < 3; ++pq)
                            dst[c][pq] = __builtin_bit_cast(
                                gbf16x8, LOAD_HANDOFF(rs_x, (i < KBW && kb < nkb) ? ((bt * nkb + kb) * 3 + pq) * 1024 + lane * 16
                                                                                  : OOB_OFFSET));
                    }
                };
                fetch(0, bf[0]);
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    if (st + 1 < NST) fetch(st + 1, bf[(st + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);                 // loads out before the MFMAs
                    const int bt = st / NCH, i0 = (st % NCH) * CH;
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const int i = i0 + c;
                        if (i < KBW) {
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[i][1], bf[st & 1][c][1], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[i][0], bf[st & 1][c][2], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[i][2], bf[st & 1][c][0], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[i][0], bf[st & 1][c][1], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[i][1], bf[st & 1][c][0], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[i][0], bf[st & 1][c][0], acc, 0, 0, 0);
                        }
                    }
                    if ((st % NCH) == NCH - 1) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) red[wave][bt][4 * q + rr][m] = acc[rr];
                        acc = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
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
            sv_r = dr_pre;
            sv_z = dz_pre;
            sv_n = dn_pre;
            sv_g = dn_pre * r;
        }
        {   // hand-off: k index of unit j in gate g is g * H + j (H % 32 == 0: an octet of units is an octet of k)
            const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(my_ring, 0, 2 * slot_bytes, 0x00020000);
            const int sbase = (s & 1) * slot_bytes + gbt * nkb * 3 * 1024 + nn * 16;
            const bool storer = gate_ok && (jj & 7) == 0;
            auto hand_off = [&](float v, int g3) {                     // (one gate at a time: 12 registers, not 36)
                u32x4 pl[3];
                split_gather_octet(v, pl);
                if (storer) {
                    const int k = g3 * H + gj;
                    const int base = sbase + (k >> 5) * 3 * 1024 + ((k & 31) >> 3) * 256;
#pragma unroll
                    for (int c = 0; c < 3; ++c) store_sc1_b128(rs_w, base + c * 1024, pl[c]);
                }
            };
            hand_off(sv_r, 0);
            hand_off(sv_z, 1);
            hand_off(sv_g, 2);
        }
        if (!DS2_DBG(dbg, 4)) wait_vmcnt0();
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gate_ok) {   // d(gi), d(gh_n) for the GEMMs that follow this launch: plain stores, off the critical path
            G[gbase] = sv_r;
            G[gbase + H] = sv_z;
            G[gbase + 2 * H] = sv_n;
            ghn[row * H + gj] = sv_g;
        }
    }
    if (tid == 0) leave_kernel(sync);
}


template <int NBT>
bool launch_fwd_persistent(float* G, float* ghn, float* hout, const float* w_hh, SyncWs* sync, float* ring, int T, int B,
                           int H, int dbg, hipStream_t st) {
    const int opts[] = {1, 2, 4, 7};
    const int kbw = pick_kbw(ds2_cdiv(H / 16, NWP), opts, 4);
    dim3 grid(ds2_cdiv(H, PJU), 2), block(NWP * 64);
#define DS2_FWD_CASE(K)                                                                                              \
    case K:                                                                                                          \
        if (!grid_is_coresident(&gru_fwd_persistent_kernel<NBT, K>, grid, 0)) return false;                          \
        hipLaunchKernelGGL((gru_fwd_persistent_kernel<NBT, K>), grid, block, 0, st, G, ghn, hout, w_hh, sync, ring, T, B,  \
                           H, dbg);                                                                                   \
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
bool launch_fwd_persistent_p2(float* G, float* ghn, float* hout, const float* w_hh, SyncWs* sync, float* ring, int T, int B,
                              int H, int dbg, hipStream_t st) {
    const int opts[] = {1, 2, 4, 7};
    const int kbw = pick_kbw(ds2_cdiv(H / 16, NWP), opts, 4);
    dim3 grid(ds2_cdiv(H, 16), 2, 2), block(NWP * 64);
#define DS2_FWDP2_CASE(K)                                                                                          \
    case K:                                                                                                        \
        if (!grid_is_coresident(&gru_fwd_persistent_p2_kernel<NBT, K>, grid, 0)) return false;                     \
        hipLaunchKernelGGL((gru_fwd_persistent_p2_kernel<NBT, K>), grid, block, 0, st, G, ghn, hout, w_hh, sync, ring,   \
                           T, B, H, dbg);                                                                          \
        return true;
    switch (kbw) {
        DS2_FWDP2_CASE(1)
        DS2_FWDP2_CASE(2)
        DS2_FWDP2_CASE(4)
        DS2_FWDP2_CASE(7)
    }
#undef DS2_FWDP2_CASE
    return false;
}

template <int NBT>
bool launch_fwd_persistent_p2b(float* G, float* ghn, float* hout, const float* w_hh, SyncWs* sync, float* ring, int T, int B,
                               int H, int dbg, hipStream_t st) {
    const int opts[] = {1, 2, 4};
    const int kbw = pick_kbw(ds2_cdiv(H / 32, NWP), opts, 3);
    dim3 grid(ds2_cdiv(H, 16), 2, 2), block(NWP * 64);
#define DS2_FWDP2B_CASE(K)                                                                                         \
    case K:                                                                                                        \
        if (!grid_is_coresident(&gru_fwd_persistent_p2b_kernel<NBT, K>, grid, 0)) return false;                    \
        hipLaunchKernelGGL((gru_fwd_persistent_p2b_kernel<NBT, K>), grid, block, 0, st, G, ghn, hout, w_hh, sync, ring,  \
                           T, B, H, dbg);                                                                          \
        return true;
    switch (kbw) {
        DS2_FWDP2B_CASE(1)
        DS2_FWDP2B_CASE(2)
        DS2_FWDP2B_CASE(4)
    }
#undef DS2_FWDP2B_CASE
    return false;
}

template <int NBT>
bool launch_bwd_persistent(float* G, float* ghn, const float* hout, const float* d_out, const float* w_hh_t,
                           SyncWs* sync, float* ring, int T, int B, int H, int dbg, hipStream_t st) {
    const int opts[] = {1, 2, 4, 8, 19};
    const int kbw = pick_kbw(ds2_cdiv(3 * H / 16, NWP), opts, 5);
    dim3 grid(ds2_cdiv(H, PJU), 2), block(NWP * 64);
#define DS2_BWD_CASE(K)                                                                                          \
    case K:                                                                                                      \
        if (!grid_is_coresident(&gru_bwd_persistent_kernel<NBT, K>, grid, 0)) return false;                      \
        hipLaunchKernelGGL((gru_bwd_persistent_kernel<NBT, K>), grid, block, 0, st, G, ghn, hout, d_out, w_hh_t, \
                           sync, ring, T, B, H, dbg);                                                            \
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

template <int NBT>
bool launch_bwd_persistent_p2(float* G, float* ghn, const float* hout, const float* d_out, const float* w_hh_t,
                              SyncWs* sync, float* ring, int T, int B, int H, int dbg, hipStream_t st) {
    const int opts[] = {1, 2, 4, 8, 19};
    const int kbw = pick_kbw(ds2_cdiv(3 * H / 16, NWP), opts, 5);
    dim3 grid(ds2_cdiv(H, 16), 2, 2), block(NWP * 64);
#define DS2_BWDP2_CASE(K)                                                                                          \
    case K:                                                                                                        \
        if (!grid_is_coresident(&gru_bwd_persistent_p2_kernel<NBT, K>, grid, 0)) return false;                     \
        hipLaunchKernelGGL((gru_bwd_persistent_p2_kernel<NBT, K>), grid, block, 0, st, G, ghn, hout, d_out, w_hh_t, \
                           sync, ring, T, B, H, dbg);                                                              \
        return true;
    switch (kbw) {
        DS2_BWDP2_CASE(1)
        DS2_BWDP2_CASE(2)
        DS2_BWDP2_CASE(4)
        DS2_BWDP2_CASE(8)
        DS2_BWDP2_CASE(19)
    }
#undef DS2_BWDP2_CASE
    return false;
}

template <int NBT>
bool launch_bwd_persistent_p2b(float* G, float* ghn, const float* hout, const float* d_out, const float* w_hh_t,
                               SyncWs* sync, float* ring, int T, int B, int H, int dbg, hipStream_t st) {
    const int opts[] = {1, 2, 5, 10};
    const int kbw = pick_kbw(ds2_cdiv(3 * H / 32, NWP), opts, 4);
    dim3 grid(ds2_cdiv(H, 16), 2, 2), block(NWP * 64);
#define DS2_BWDP2B_CASE(K)                                                                                          \
    case K:                                                                                                         \
        if (!grid_is_coresident(&gru_bwd_persistent_p2b_kernel<NBT, K>, grid, 0)) return false;                     \
        hipLaunchKernelGGL((gru_bwd_persistent_p2b_kernel<NBT, K>), grid, block, 0, st, G, ghn, hout, d_out, w_hh_t, \
                           sync, ring, T, B, H, dbg);                                                               \
        return true;
    switch (kbw) {
        DS2_BWDP2B_CASE(1)
        DS2_BWDP2B_CASE(2)
        DS2_BWDP2B_CASE(5)
        DS2_BWDP2B_CASE(10)
    }
#undef DS2_BWDP2B_CASE
    return false;
}

}  // namespace

bool ds2_p16_launch_fwd(int form, int nbt, float* G, float* ghn, float* hout, const float* w_hh, void* sync_, float* ring, int T,
                        int B, int H, int dbg, hipStream_t st) {
    SyncWs* sync = (SyncWs*)sync_;
#define DS2_P16_F(FN, N) FN<N>(G, ghn, hout, w_hh, sync, ring, T, B, H, dbg, st)
    if (form == 0) return nbt == 1 ? DS2_P16_F(launch_fwd_persistent, 1) : (nbt == 2 ? DS2_P16_F(launch_fwd_persistent, 2) : DS2_P16_F(launch_fwd_persistent, 4));
    if (form == 1) return nbt == 1 ? DS2_P16_F(launch_fwd_persistent_p2, 1) : DS2_P16_F(launch_fwd_persistent_p2, 2);
    return nbt == 1 ? DS2_P16_F(launch_fwd_persistent_p2b, 1) : DS2_P16_F(launch_fwd_persistent_p2b, 2);
#undef DS2_P16_F
}

bool ds2_p16_launch_bwd(int form, int nbt, float* G, float* ghn, const float* hout, const float* d_out, const float* w_hh_t,
                        void* sync_, float* ring, int T, int B, int H, int dbg, hipStream_t st) {
    SyncWs* sync = (SyncWs*)sync_;
#define DS2_P16_B(FN, N) FN<N>(G, ghn, hout, d_out, w_hh_t, sync, ring, T, B, H, dbg, st)
    if (form == 0) return nbt == 1 ? DS2_P16_B(launch_bwd_persistent, 1) : (nbt == 2 ? DS2_P16_B(launch_bwd_persistent, 2) : DS2_P16_B(launch_bwd_persistent, 4));
    if (form == 1) return nbt == 1 ? DS2_P16_B(launch_bwd_persistent_p2, 1) : DS2_P16_B(launch_bwd_persistent_p2, 2);
    return nbt == 1 ? DS2_P16_B(launch_bwd_persistent_p2b, 1) : DS2_P16_B(launch_bwd_persistent_p2b, 2);
#undef DS2_P16_B
}

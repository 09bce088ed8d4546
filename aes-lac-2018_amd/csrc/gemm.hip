// fp32 GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32) for gfx950.
//
// C[M,N] = op(A) op(B) + beta C.  One 256-thread workgroup (4 waves, 2x2) computes a 128x128
// tile; each wave owns 64x64 = 2x2 MFMA tiles of 32x32 (4 accumulators x 16 VGPRs).  K is
// walked in 16-deep slabs staged through LDS in a k-major image ([k][m], [k][n]) so that
// the MFMA operand fetch (lane l: A[m = l&31][k = l>>5]) is a conflict-free ds_read_b32 of
// 32 consecutive floats per half-wave.  Global->register prefetch of slab k+1 overlaps the
// MFMAs of slab k (double-buffered LDS, one barrier per slab).
//
// Operand source layouts (row-major, leading dimension ld):
//   A "k-contiguous"  A[m*lda + k]   (trans_a = 0)     A "m-contiguous"  A[k*lda + m]  (trans_a = 1)
//   B "n-contiguous"  B[k*ldb + n]   (trans_b = 0)     B "k-contiguous"  B[n*ldb + k]  (trans_b = 1)
// Numerics: exact fp32 FMA chain in k order per 2-k MFMA (same as fmaf accumulation).
#include <stdlib.h>

#include <type_traits>

#include "ds2_common.h"
#include "split_bf16.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int NI = BK / 8;  // float4 per thread per operand slab
constexpr int LDT = 132;  // LDS row pitch (floats): 16-B aligned rows, 2-way conflict at worst on writes

// Stage one operand slab [BK][128] into registers.  KCONTIG: the source is X x K (x = m or n).
// Every load is an UNCONDITIONAL buffer load: an element outside the tile's valid range gets an offset beyond the
// descriptor and the hardware range check returns 0.  (A guarded or selected load makes hipcc emit a branch and a
// vmcnt(0) per load -- 320 branches in this kernel before the change.)
constexpr int OOB = 0x7FFFFFF0;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t rs, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t rs, int byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, 0));
}

template <bool KCONTIG, bool VEC>
__device__ __forceinline__ void load_slab(__amdgpu_buffer_rsrc_t rs, int ld, int x0, int xmax, int k0, int kmax,
                                          int tid, f32x4 (&r)[NI]) {
    if (KCONTIG) {
        const int kq = (tid & (BK / 4 - 1)) * 4;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int x = x0 + tid / (BK / 4) + (1024 / BK) * i;
            const int k = k0 + kq;
            const int off = (x * ld + k) * 4;
            if (VEC) {
                r[i] = bload4(rs, (x < xmax && k + 3 < kmax) ? off : OOB);
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) r[i][c] = bload1(rs, (x < xmax && k + c < kmax) ? off + 4 * c : OOB);
            }
        }
    } else {
        const int xq = (tid & 31) * 4;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int k = k0 + (tid >> 5) + 8 * i;
            const int x = x0 + xq;
            const int off = (k * ld + x) * 4;
            if (VEC) {
                if (x + 3 < xmax || x >= xmax) {          // whole chunk in or out (the common case, no per-element work)
                    r[i] = bload4(rs, (k < kmax && x < xmax) ? off : OOB);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) r[i][c] = bload1(rs, (k < kmax && x + c < xmax) ? off + 4 * c : OOB);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) r[i][c] = bload1(rs, (k < kmax && x + c < xmax) ? off + 4 * c : OOB);
            }
        }
    }
}

// Fast path of load_slab for whole slabs: the per-lane byte offsets of a tile do not change from slab to slab, only a
// wave-uniform advance does -- it goes into the buffer instruction's SCALAR offset.  Lanes outside the tile keep an
// out-of-range vector offset (the range check ignores the scalar offset), so the loop body holds no address arithmetic:
// the general form above recomputes and re-checks every offset for every slab, ~80 vector instructions in front of each
// slab's first MFMA.  Valid when every 16-byte chunk is wholly inside or outside the tile (VEC and xmax % 4 == 0 for the
// x-contiguous form) and the slab does not cross kmax.
template <bool KCONTIG>
__device__ __forceinline__ void slab_voffsets(int ld, int x0, int xmax, int tid, int (&voff)[NI]) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (KCONTIG) {
            const int x = x0 + tid / (BK / 4) + (1024 / BK) * i;
            voff[i] = x < xmax ? (x * ld + (tid & (BK / 4 - 1)) * 4) * 4 : OOB;
        } else {
            const int x = x0 + (tid & 31) * 4;
            voff[i] = x < xmax ? (((tid >> 5) + 8 * i) * ld + x) * 4 : OOB;
        }
    }
}
template <bool KCONTIG>
__device__ __forceinline__ int slab_soffset(int ld, int k0) {
    return KCONTIG ? k0 * 4 : k0 * ld * 4;
}
__device__ __forceinline__ void load_slab_fast(__amdgpu_buffer_rsrc_t rs, const int (&voff)[NI], int soff, f32x4 (&r)[NI]) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
        r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], soff, 0));
}

template <bool KCONTIG>
__device__ __forceinline__ void store_slab(float* __restrict__ lds, int tid, const f32x4 (&r)[NI]) {
    if (KCONTIG) {
        const int kq = (tid & (BK / 4 - 1)) * 4;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int x = tid / (BK / 4) + (1024 / BK) * i;
#pragma unroll
            for (int c = 0; c < 4; ++c) lds[(kq + c) * LDT + x] = r[i][c];
        }
    } else {
        const int xq = (tid & 31) * 4;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int k = (tid >> 5) + 8 * i;
            *reinterpret_cast<f32x4*>(&lds[k * LDT + xq]) = r[i];
        }
    }
}

// XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, so workgroup b and b+8 share an L2.
// Give every XCD a contiguous run of tiles (neighbours share an A panel) -- bijective for any tile count.
__device__ __forceinline__ int xcd_tile(int wg, int nwg) {
    const int xcd = wg & 7, idx = wg >> 3;
    const int qd = nwg >> 3, rm = nwg & 7;
    return (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
}

// one 128 x 128 tile over the k range of one split (the body of the single and of the grouped launch).
// MODE 0: the four waves sit 2 x 2, 64 x 64 outputs = 4 MFMAs per k pair each.  MODE 1 / 2: an edge tile with at most 32 / 64
// valid columns (N = 800 = 6 x 128 + 32: every seventh tile of dX, dW_ih and dW_hh): the waves stack in M (wave w = rows
// 32 w .. 32 w + 31) and issue 1 / 2 MFMAs per k pair -- the tile then costs a quarter / half of the matrix-pipe time instead
// of all of it for columns that do not exist (12 % of those GEMMs).  The choice is per workgroup, outside the slab loop.
template <bool A_KCONTIG, bool B_KCONTIG, bool VEC, int MODE>
__device__ __forceinline__ void gemm_tile_body(float (&lds)[2][2][BK * LDT], int M, int N, const float* __restrict__ A, int lda,
                                               const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
                                               float beta, int use_atomic, unsigned int a_bytes, unsigned int b_bytes, int m0,
                                               int n0, int kbeg, int kend) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    constexpr int NI_ = MODE == 0 ? 2 : 1, NJ_ = MODE == 0 ? 2 : MODE;
    const int arow = MODE == 0 ? wm * 64 : wave * 32, bcol = MODE == 0 ? wn * 64 : 0;

    f32x16 acc[NI_][NJ_];
#pragma unroll
    for (int i = 0; i < NI_; ++i)
#pragma unroll
        for (int j = 0; j < NJ_; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, b_bytes, 0x00020000);
    f32x4 ra[NI], rb[NI];
    const int nslab = (kend - kbeg + BK - 1) / BK;
    if (nslab > 0) {
        load_slab<A_KCONTIG, VEC>(rsa, lda, m0, M, kbeg, kend, tid, ra);
        load_slab<B_KCONTIG, VEC>(rsb, ldb, n0, N, kbeg, kend, tid, rb);
        store_slab<A_KCONTIG>(lds[0][0], tid, ra);
        store_slab<B_KCONTIG>(lds[0][1], tid, rb);
    }
    __syncthreads();

    const int lr = lane & 31, lh = lane >> 5;
    for (int s = 0; s < nslab; ++s) {
        const int cur = s & 1;
#ifndef DS2_GEMM_ABL
#define DS2_GEMM_ABL 0      // timing-only ablations (results WRONG): 1 no LDS stores, 2 no global loads, 4 no LDS reads
#endif
        if (s + 1 < nslab && !(DS2_GEMM_ABL & 2)) {
            load_slab<A_KCONTIG, VEC>(rsa, lda, m0, M, kbeg + (s + 1) * BK, kend, tid, ra);
            load_slab<B_KCONTIG, VEC>(rsb, ldb, n0, N, kbeg + (s + 1) * BK, kend, tid, rb);
        }
        const float* as = lds[cur][0] + arow + lr;
        const float* bs = lds[cur][1] + bcol + lr;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[NI_], b[NJ_];
#pragma unroll
            for (int i = 0; i < NI_; ++i) a[i] = (DS2_GEMM_ABL & 4) ? ra[0][0] + kk : as[(kk + lh) * LDT + 32 * i];
#pragma unroll
            for (int j = 0; j < NJ_; ++j) b[j] = (DS2_GEMM_ABL & 4) ? rb[0][0] + kk : bs[(kk + lh) * LDT + 32 * j];
#pragma unroll
            for (int i = 0; i < NI_; ++i)
#pragma unroll
                for (int j = 0; j < NJ_; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nslab && !(DS2_GEMM_ABL & 1)) {
            store_slab<A_KCONTIG>(lds[cur ^ 1][0], tid, ra);
            store_slab<B_KCONTIG>(lds[cur ^ 1][1], tid, rb);
        }
        __syncthreads();
    }

    // C/D map of 32x32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < NI_; ++i)
#pragma unroll
        for (int j = 0; j < NJ_; ++j) {
            const int n = n0 + bcol + j * 32 + lr;
            if (n >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + arow + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M) {
                    float* c = C + (size_t)m * ldc + n;
                    if (use_atomic)
                        atomicAdd(c, acc[i][j][r]);
                    else
                        *c = (beta != 0.f) ? acc[i][j][r] + beta * (*c) : acc[i][j][r];
                }
            }
        }
}

template <bool A_KCONTIG, bool B_KCONTIG, bool VEC>
__device__ __forceinline__ void gemm_tile(int M, int N, int K, const float* __restrict__ A, int lda,
                                          const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc, float beta,
                                          int tiles_n, int k_per_split, int use_atomic, unsigned int a_bytes,
                                          unsigned int b_bytes, int tile, int split) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][BK * LDT];  // [buf][A|B]
    const int m0 = (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * BN;
    const int kbeg = split * k_per_split;
    const int kend = min(K, kbeg + k_per_split);
    const int ncols = N - n0;                                            // wave-uniform (scalar) choice
    if (ncols <= 32)
        gemm_tile_body<A_KCONTIG, B_KCONTIG, VEC, 1>(lds, M, N, A, lda, B, ldb, C, ldc, beta, use_atomic, a_bytes, b_bytes, m0,
                                                     n0, kbeg, kend);
    else if (ncols <= 64)
        gemm_tile_body<A_KCONTIG, B_KCONTIG, VEC, 2>(lds, M, N, A, lda, B, ldb, C, ldc, beta, use_atomic, a_bytes, b_bytes, m0,
                                                     n0, kbeg, kend);
    else
        gemm_tile_body<A_KCONTIG, B_KCONTIG, VEC, 0>(lds, M, N, A, lda, B, ldb, C, ldc, beta, use_atomic, a_bytes, b_bytes, m0,
                                                     n0, kbeg, kend);
}

template <bool A_KCONTIG, bool B_KCONTIG, bool VEC>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(int M, int N, int K, const float* __restrict__ A,
                                                          int lda, const float* __restrict__ B, int ldb,
                                                          float* __restrict__ C, int ldc, float beta,
                                                          int tiles_n, int k_per_split, int use_atomic,
                                                          unsigned int a_bytes, unsigned int b_bytes) {
    gemm_tile<A_KCONTIG, B_KCONTIG, VEC>(M, N, K, A, lda, B, ldb, C, ldc, beta, tiles_n, k_per_split, use_atomic, a_bytes,
                                         b_bytes, xcd_tile(blockIdx.x, gridDim.x), blockIdx.y);
}

// Up to four independent TN problems that share K and N in ONE launch (blockIdx.z = problem): the four weight-gradient
// GEMMs dW_hh of a BiGRU layer (two directions x {r|z rows, n rows}) are 1600 x 800 and 800 x 800 outputs -- 91 and 49
// tiles -- and ran at 74 / 58 TFLOP/s as separate launches; together they fill the chip like the 4800 x 800 dW_ih does.
struct GemmGroup {
    const float* A[4];
    const float* B[4];
    float* C[4];
    int M[4], lda[4], ldb[4], ldc[4];
    unsigned int a_bytes[4], b_bytes[4];
};
template <bool VEC>
__global__ __launch_bounds__(256, 2) void gemm_f32_tn_group_kernel(GemmGroup g, int N, int K, int tiles_n, int k_per_split) {
    const int p = blockIdx.z;
    const int tiles = ((g.M[p] + BM - 1) / BM) * tiles_n;
    if ((int)blockIdx.x >= tiles) return;
    gemm_tile<false, false, VEC>(g.M[p], N, K, g.A[p], g.lda[p], g.B[p], g.ldb[p], g.C[p], g.ldc[p], 0.f, tiles_n, k_per_split,
                                 1, g.a_bytes[p], g.b_bytes[p], blockIdx.x, blockIdx.y);
}

// ----------------------------------------------------------------------------------------------------------
// v2 main loop: THREE LDS slab buffers.  In iteration s the MFMAs read slab s, the registers loaded one iteration ago
// (slab s+2) are written to the third buffer, the global loads of slab s+3 are issued, and the first operands of slab
// s+1 -- written an iteration ago, already behind a barrier -- are fetched BEFORE this iteration's barrier, so the
// first MFMA after the barrier has its operands and no wave waits on a global load it issued moments ago.  (The
// two-buffer loop above stalls every 16-deep slab on the vmcnt wait + LDS store + barrier + first ds_read chain:
// SQ_WAIT_ANY 16 % of wave cycles, MFMA pipe ~75 % busy on 4096^3.)
//
// Work decomposition: a workgroup runs one or two PIECES = (tile, slab range).  Data-parallel pieces own a whole tile
// and store it; when the tile count is not a multiple of the resident slots (2 per CU), the tiles of the last, partial
// round are cut "stream-K" style into equal runs of (tile, slab) units -- one run per slot, crossing at most one tile
// boundary -- and accumulated with float atomics into rows the launcher zeroed: 1292 tiles on 512 slots take
// 2.54 tile-times instead of 3.
// ----------------------------------------------------------------------------------------------------------
template <bool A_KCONTIG, bool B_KCONTIG, bool VEC>
__global__ __launch_bounds__(256, 2) void gemm_f32_v2_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                             const float* __restrict__ B, int ldb,
                                                             float* __restrict__ C, int ldc, float beta, int tiles_n,
                                                             int dp_tiles, int sk_units, int sk_total,
                                                             unsigned int a_bytes, unsigned int b_bytes) {
    __shared__ __attribute__((aligned(16))) float lds[3][2][BK * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int nslab_all = (K + BK - 1) / BK;
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, b_bytes, 0x00020000);

    int u0 = 0, u1 = 0;                                // stream-K unit range of this workgroup (empty for DP workgroups)
    const bool is_dp = (int)blockIdx.x < dp_tiles;
    if (!is_dp) {
        u0 = ((int)blockIdx.x - dp_tiles) * sk_units;
        u1 = min(u0 + sk_units, sk_total);
    }
#pragma unroll 1
    for (int piece = 0; piece < 2; ++piece) {
        int tile, s_beg, s_end;
        if (is_dp) {
            if (piece == 1) break;
            // XCD-aware order over the data-parallel tiles (workgroups b and b + 8 share an L2)
            const int nwg = dp_tiles, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
            const int qd = nwg >> 3, rm = nwg & 7;
            tile = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
            s_beg = 0;
            s_end = nslab_all;
        } else {
            if (u0 >= u1) break;
            tile = dp_tiles + u0 / nslab_all;
            s_beg = u0 % nslab_all;
            s_end = min(nslab_all, s_beg + (u1 - u0));
            u0 += s_end - s_beg;
        }
        const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
        const int nslab = s_end - s_beg;
        const int kbeg = s_beg * BK, kend = min(K, s_end * BK);

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        f32x4 ra[NI], rb[NI];
        int va[NI], vb[NI];
        slab_voffsets<A_KCONTIG>(lda, m0, M, tid, va);
        slab_voffsets<B_KCONTIG>(ldb, n0, N, tid, vb);
        // whole slabs of a tile whose 16-byte chunks never straddle its edge take the fast loads
        const bool fast_ok = VEC && (A_KCONTIG || (M & 3) == 0) && (B_KCONTIG || (N & 3) == 0);
        auto fetch = [&](int k0) {
            if (fast_ok && k0 + BK <= kend) {
                load_slab_fast(rsa, va, slab_soffset<A_KCONTIG>(lda, k0), ra);
                load_slab_fast(rsb, vb, slab_soffset<B_KCONTIG>(ldb, k0), rb);
            } else {
                load_slab<A_KCONTIG, VEC>(rsa, lda, m0, M, k0, kend, tid, ra);
                load_slab<B_KCONTIG, VEC>(rsb, ldb, n0, N, k0, kend, tid, rb);
            }
        };
        __syncthreads();                               // (second piece: the previous piece's LDS reads are done)
        fetch(kbeg);
        store_slab<A_KCONTIG>(lds[0][0], tid, ra);
        store_slab<B_KCONTIG>(lds[0][1], tid, rb);
        if (nslab > 1) {
            fetch(kbeg + BK);
            store_slab<A_KCONTIG>(lds[1][0], tid, ra);
            store_slab<B_KCONTIG>(lds[1][1], tid, rb);
        }
        if (nslab > 2) {                               // slab 2 stays in registers until iteration 0 stores it
            fetch(kbeg + 2 * BK);
        }
        __syncthreads();
        const int aoff = wm * 64 + lr + lh * LDT, boff = wn * 64 + lr + lh * LDT;
        float pa0 = lds[0][0][aoff], pa1 = lds[0][0][aoff + 32], pb0 = lds[0][1][boff], pb1 = lds[0][1][boff + 32];
        int cur = 0;
        for (int s = 0; s < nslab; ++s) {
            const float* as = lds[cur][0] + aoff;
            const float* bs = lds[cur][1] + boff;
            const int nxt = cur == 2 ? 0 : cur + 1, nx2 = nxt == 2 ? 0 : nxt + 1;
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const float a0 = pa0, a1 = pa1, b0 = pb0, b1 = pb1;
                if (kk + 2 < BK) {                     // next k pair of this slab
                    pa0 = as[(kk + 2) * LDT];
                    pa1 = as[(kk + 2) * LDT + 32];
                    pb0 = bs[(kk + 2) * LDT];
                    pb1 = bs[(kk + 2) * LDT + 32];
                }
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                // pin the order "operand reads of the NEXT k pair, then the four MFMAs of this one": left alone the
                // scheduler sinks the reads behind the MFMAs, right in front of their wait, and every k pair exposes an
                // LDS round trip
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     // DS reads
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);     // MFMA
                if (kk == BK - 6 && s + 2 < nslab) {   // registers (slab s+2, loaded an iteration ago) -> third buffer
                    store_slab<A_KCONTIG>(lds[nx2][0], tid, ra);
                    store_slab<B_KCONTIG>(lds[nx2][1], tid, rb);
                    if (s + 3 < nslab) fetch(kbeg + (s + 3) * BK);
                }
            }
            if (s + 1 < nslab) {                       // first operands of slab s+1: its buffer is a barrier old
                pa0 = lds[nxt][0][aoff];
                pa1 = lds[nxt][0][aoff + 32];
                pb0 = lds[nxt][1][boff];
                pb1 = lds[nxt][1][boff + 32];
            }
            __syncthreads();
            cur = nxt;
        }

        const bool atomic = !is_dp;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + lr;
                if (n >= N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m < M) {
                        float* c = C + (size_t)m * ldc + n;
                        if (atomic)
                            atomicAdd(c, acc[i][j][r]);
                        else
                            *c = (beta != 0.f) ? acc[i][j][r] + beta * (*c) : acc[i][j][r];
                    }
                }
            }
    }
}

// ----------------------------------------------------------------------------------------------------------
// fp32 GEMM on the bf16 matrix pipe: error-free operand splitting.
//
// Every fp32 operand element is written as a = a1 + a2 + a3 with a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)
// (round-to-nearest-even at each step; bf16 has fp32's exponent range and 8 significant bits, so the three terms carry all
// 24 bits of a: |a2| <= 2^-8 |a|, |a3| <= 2^-16 |a|, a - a1 - a2 - a3 = 0 barring underflow).  A product a b is then the sum
// of nine partial products ai bj, each EXACT in the pipe's fp32 accumulator (8 x 8 significant bits).  NPROD = 9 adds all
// of them; NPROD = 6 leaves out a2 b3, a3 b2 and a3 b3, together below the rounding error of ONE
// fp32 product, and far below what the fp32 accumulation of K such products leaves in either kernel family.  The
// accumulator is fp32 as in the f32-input MFMA kernels above.  v_mfma_f32_32x32x16_bf16 runs 16x the f32-input MFMA's
// rate, so six of them per 16 k are 2.7x the f32 pipe's peak for the same fp32 result (tests/test_kernels_gpu.py checks
// both families against an fp64 product; tools/gemm_bench.py times them).
//
// Same 128 x 128 tile / 2 x 2 waves / 16-deep slab / double-buffered LDS structure as gemm_tile_body; the split happens
// once per element on the way from the prefetch registers into LDS (5.5 vector instructions per element), the LDS image
// is [row][plane][16 k] bf16 (row pitch 112 B: the 16-byte fragment reads of a 32-row tile are conflict-free) and a
// wave's twelve fragments (2 row tiles + 2 column tiles, 3 planes each) feed 24 MFMAs.
// ----------------------------------------------------------------------------------------------------------
constexpr int SP = SPLIT_PITCH;                 // LDS row pitch, bytes: 3 planes x 16 k x 2 B + 16 (split_bf16.h)
constexpr int SPLIT_TILE_BYTES = 128 * SP;      // one operand slab

// Prefetch registers of one operand slab: 8 floats per thread.
//   k-contiguous source: r[4 i + c] = element (x = tid / 4 + 64 i, k = 4 (tid & 3) + c)        (two 16-byte loads)
//   x-contiguous source: r[c]       = element (x = tid & 127,      k = 8 (tid >> 7) + c)        (eight 4-byte loads, each
//                                     coalesced over the wave's 64 consecutive x) -- the thread then owns 8 consecutive k
//                                     of one row, i.e. whole 16-byte runs of the [row][k] image: the transposition costs
//                                     nothing
// The per-lane byte offsets are computed ONCE per tile (rows outside the tile get an offset the descriptor's range check
// rejects: the load returns 0); the slab's advance is wave-uniform and rides in the instruction's scalar offset, so the
// slab loop holds no vector address arithmetic.  k beyond the operand: an x-contiguous source runs off the end of the
// descriptor (k * ld + x >= (K - 1) * ld + xmax) and reads 0 by itself; a k-contiguous one would read its next row, so
// the launcher only selects these kernels for it when K is a multiple of the slab depth.
template <bool KCONTIG>
__device__ __forceinline__ void split_voffsets(int ld, int x0, int xmax, int tid, int (&voff)[2]) {
    if (KCONTIG) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int x = x0 + (tid >> 2) + 64 * i;
            voff[i] = x < xmax ? (x * ld + (tid & 3) * 4) * 4 : OOB;
        }
    } else {
        const int x = x0 + (tid & 127);
        voff[0] = voff[1] = x < xmax ? (8 * (tid >> 7) * ld + x) * 4 : OOB;
    }
}
template <bool KCONTIG>
__device__ __forceinline__ void split_load(__amdgpu_buffer_rsrc_t rs, const int (&voff)[2], int ld, int k0, float (&r)[8]) {
    if (KCONTIG) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], k0 * 4, 0));
#pragma unroll
            for (int c = 0; c < 4; ++c) r[4 * i + c] = v[c];
        }
    } else {
#pragma unroll
        for (int c = 0; c < 8; ++c)
            r[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[0], (k0 + c) * ld * 4, 0));
    }
}
template <bool KCONTIG>
__device__ __forceinline__ void split_store(char* __restrict__ lds, int tid, const float (&r)[8]) {
    if (KCONTIG) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned int p[3][2];
            split3(r[4 * i], r[4 * i + 1], p[0][0], p[1][0], p[2][0]);
            split3(r[4 * i + 2], r[4 * i + 3], p[0][1], p[1][1], p[2][1]);
            char* row = lds + ((tid >> 2) + 64 * i) * SP + (tid & 3) * 8;
#pragma unroll
            for (int q = 0; q < 3; ++q) *reinterpret_cast<uint2*>(row + 32 * q) = make_uint2(p[q][0], p[q][1]);
        }
    } else {
        unsigned int p[3][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) split3(r[2 * c], r[2 * c + 1], p[0][c], p[1][c], p[2][c]);
        char* row = lds + (tid & 127) * SP + (tid >> 7) * 16;
#pragma unroll
        for (int q = 0; q < 3; ++q) *reinterpret_cast<uint4*>(row + 32 * q) = make_uint4(p[q][0], p[q][1], p[q][2], p[q][3]);
    }
}

template <bool A_KCONTIG, bool B_KCONTIG, int MODE, int NPROD>
__device__ __forceinline__ void gemm_split_tile_body(char (&lds)[2][2][SPLIT_TILE_BYTES], int M, int N,
                                                     const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                     int ldb, float* __restrict__ C, int ldc, float beta, int use_atomic,
                                                     unsigned int a_bytes, unsigned int b_bytes, int m0, int n0, int kbeg,
                                                     int kend) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    constexpr int NI_ = MODE == 0 ? 2 : 1, NJ_ = MODE == 0 ? 2 : MODE;
    const int arow = MODE == 0 ? wm * 64 : wave * 32, bcol = MODE == 0 ? wn * 64 : 0;

    f32x16 acc[NI_][NJ_], acc_lo[NI_][NJ_];           // leading products / products with a residual term (split_mfma2)
#pragma unroll
    for (int i = 0; i < NI_; ++i)
#pragma unroll
        for (int j = 0; j < NJ_; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = acc_lo[i][j][r] = 0.f;

    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, b_bytes, 0x00020000);
    // Two register sets: the loads of slab s + 2 are issued while slab s is multiplied and slab s + 1 (loaded one
    // iteration earlier) is split into the other LDS buffer -- a 16-deep slab is only 24 MFMAs = 768 pipe cycles per wave,
    // far less than a global load's latency, so one slab of prefetch distance left every wave parked on vmcnt.
    float ra[2][8], rb[2][8];
    int va[2], vb[2];
    split_voffsets<A_KCONTIG>(lda, m0, M, tid, va);
    split_voffsets<B_KCONTIG>(ldb, n0, N, tid, vb);
    const int nslab = (kend - kbeg + BK - 1) / BK;
    if (nslab > 0) {
        split_load<A_KCONTIG>(rsa, va, lda, kbeg, ra[0]);
        split_load<B_KCONTIG>(rsb, vb, ldb, kbeg, rb[0]);
    }
    if (nslab > 1) {
        split_load<A_KCONTIG>(rsa, va, lda, kbeg + BK, ra[1]);
        split_load<B_KCONTIG>(rsb, vb, ldb, kbeg + BK, rb[1]);
    }
    if (nslab > 0) {
        split_store<A_KCONTIG>(lds[0][0], tid, ra[0]);
        split_store<B_KCONTIG>(lds[0][1], tid, rb[0]);
    }
    __syncthreads();

    const int lr = lane & 31, lh = lane >> 5;
    const int a_off = (arow + lr) * SP + lh * 16, b_off = (bcol + lr) * SP + lh * 16;
    auto slab = [&](int s, auto PAR) {
        constexpr int cur = decltype(PAR)::value;                        // s & 1: LDS buffer and register set of slab s
        // (unconditional: past the last slab the loads fetch zeros or data nobody uses -- under a condition the compiler's
        // wait-count insertion has to assume they were NOT issued and waits for them together with slab s + 1's)
        if (!(DS2_GEMM_ABL & 2)) {
        split_load<A_KCONTIG>(rsa, va, lda, kbeg + (s + 2) * BK, ra[cur]);
        split_load<B_KCONTIG>(rsb, vb, ldb, kbeg + (s + 2) * BK, rb[cur]);
        }
        const char* as = lds[cur][0] + a_off;
        const char* bs = lds[cur][1] + b_off;
        bf16x8 a[NI_][3], b[NJ_][3];
#pragma unroll
        for (int i = 0; i < NI_; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                a[i][q] = (DS2_GEMM_ABL & 4) ? __builtin_bit_cast(bf16x8, f32x4{ra[0][0] + q, ra[0][1], ra[0][2], ra[0][3] + i})
                                             : *reinterpret_cast<const bf16x8*>(as + i * 32 * SP + q * 32);
#pragma unroll
        for (int j = 0; j < NJ_; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                b[j][q] = (DS2_GEMM_ABL & 4) ? __builtin_bit_cast(bf16x8, f32x4{rb[0][0] + q, rb[0][1], rb[0][2], rb[0][3] + j})
                                             : *reinterpret_cast<const bf16x8*>(bs + j * 32 * SP + q * 32);
#pragma unroll
        for (int i = 0; i < NI_; ++i)
#pragma unroll
            for (int j = 0; j < NJ_; ++j) {
                split_mfma2<NPROD>(a[i], b[j], acc[i][j], acc_lo[i][j]);
            }
        // (unconditional as well: behind the last slab it fills a buffer nobody reads)
        if (!(DS2_GEMM_ABL & 1)) {
        split_store<A_KCONTIG>(lds[cur ^ 1][0], tid, ra[cur ^ 1]);
        split_store<B_KCONTIG>(lds[cur ^ 1][1], tid, rb[cur ^ 1]);
        }
        // issue order: the compiler puts the 88 vector instructions of the split behind the last MFMA; spread them (and the
        // LDS stores) between the MFMAs, whose issue takes 8 of their 32 pipe cycles
        constexpr int NMFMA = NI_ * NJ_ * NPROD, VPM = (96 + NMFMA - 1) / NMFMA, WEVERY = NMFMA >= 8 ? NMFMA / 8 : 1;
        __builtin_amdgcn_sched_group_barrier(0x020, (A_KCONTIG ? 2 : 8) + (B_KCONTIG ? 2 : 8), 0);     // slab s + 2's loads first
        __builtin_amdgcn_sched_group_barrier(0x100, 3 * (NI_ + NJ_), 0);
#pragma unroll
        for (int g = 0; g < NMFMA; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
            if (g % WEVERY == WEVERY - 1) __builtin_amdgcn_sched_group_barrier(0x200, NMFMA >= 8 ? 1 : 8 / NMFMA + 1, 0);
        }
        __syncthreads();
    };
    for (int s = 0; s < nslab; s += 2) {
        slab(s, std::integral_constant<int, 0>{});
        if (s + 1 < nslab) slab(s + 1, std::integral_constant<int, 1>{});
    }
#pragma unroll
    for (int i = 0; i < NI_; ++i)
#pragma unroll
        for (int j = 0; j < NJ_; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += acc_lo[i][j][r];

#pragma unroll
    for (int i = 0; i < NI_; ++i)
#pragma unroll
        for (int j = 0; j < NJ_; ++j) {
            const int n = n0 + bcol + j * 32 + lr;
            if (n >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + arow + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M) {
                    float* c = C + (size_t)m * ldc + n;
                    if (use_atomic)
                        atomicAdd(c, acc[i][j][r]);
                    else
                        *c = (beta != 0.f) ? acc[i][j][r] + beta * (*c) : acc[i][j][r];
                }
            }
        }
}

template <bool A_KCONTIG, bool B_KCONTIG, int NPROD>
__device__ __forceinline__ void gemm_split_tile(int M, int N, int K, const float* __restrict__ A, int lda,
                                                const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
                                                float beta, int tiles_n, int k_per_split, int use_atomic,
                                                unsigned int a_bytes, unsigned int b_bytes, int tile, int split) {
    __shared__ __attribute__((aligned(16))) char lds[2][2][SPLIT_TILE_BYTES];  // [buf][A|B]
    const int m0 = (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * BN;
    const int kbeg = split * k_per_split;
    const int kend = min(K, kbeg + k_per_split);
    const int ncols = N - n0;                                            // wave-uniform (scalar) choice
    if (ncols <= 32)
        gemm_split_tile_body<A_KCONTIG, B_KCONTIG, 1, NPROD>(lds, M, N, A, lda, B, ldb, C, ldc, beta, use_atomic, a_bytes,
                                                             b_bytes, m0, n0, kbeg, kend);
    else if (ncols <= 64)
        gemm_split_tile_body<A_KCONTIG, B_KCONTIG, 2, NPROD>(lds, M, N, A, lda, B, ldb, C, ldc, beta, use_atomic, a_bytes,
                                                             b_bytes, m0, n0, kbeg, kend);
    else
        gemm_split_tile_body<A_KCONTIG, B_KCONTIG, 0, NPROD>(lds, M, N, A, lda, B, ldb, C, ldc, beta, use_atomic, a_bytes,
                                                             b_bytes, m0, n0, kbeg, kend);
}

template <bool A_KCONTIG, bool B_KCONTIG, int NPROD>
__global__ __launch_bounds__(256, 2) void gemm_bf16x_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                            const float* __restrict__ B, int ldb, float* __restrict__ C,
                                                            int ldc, float beta, int tiles_n, int k_per_split,
                                                            int use_atomic, unsigned int a_bytes, unsigned int b_bytes) {
    gemm_split_tile<A_KCONTIG, B_KCONTIG, NPROD>(M, N, K, A, lda, B, ldb, C, ldc, beta, tiles_n, k_per_split, use_atomic,
                                                 a_bytes, b_bytes, xcd_tile(blockIdx.x, gridDim.x), blockIdx.y);
}
template <int NPROD>
__global__ __launch_bounds__(256, 2) void gemm_bf16x_tn_group_kernel(GemmGroup g, int N, int K, int tiles_n, int k_per_split) {
    const int p = blockIdx.z;
    const int tiles = ((g.M[p] + BM - 1) / BM) * tiles_n;
    if ((int)blockIdx.x >= tiles) return;
    gemm_split_tile<false, false, NPROD>(g.M[p], N, K, g.A[p], g.lda[p], g.B[p], g.ldb[p], g.C[p], g.ldc[p], 0.f, tiles_n,
                                         k_per_split, 1, g.a_bytes[p], g.b_bytes[p], blockIdx.x, blockIdx.y);
}

// Kernel family: 0 = f32-input MFMA kernels only, 6 (default) / 9 = the split-operand kernels with that many partial
// products.  DS2_GEMM_SPLIT sets the process default; ds2_gemm_split_mode() changes it at run time (the tests compare the
// families in one process).
inline int& gemm_split_mode_ref() {
    static int mode = [] {
        const char* e = getenv("DS2_GEMM_SPLIT");
        const int v = e ? atoi(e) : 6;
        return (v == 6 || v == 9) ? v : 0;
    }();
    return mode;
}
inline int gemm_split_mode() { return gemm_split_mode_ref(); }

// zero `rows` rows of N floats with leading dimension ldc: ONE linear fill when the rows are contiguous (the 2-D fill of the
// runtime runs at ~0.4 TB/s: 38 us for the 16 MB of a dX output against ~8 us for the linear one)
inline void zero_rows(float* C, int ldc, int N, int rows, hipStream_t st) {
    if (rows <= 0) return;
    if (ldc == N)
        (void)hipMemsetAsync(C, 0, (size_t)rows * N * sizeof(float), st);
    else
        (void)hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), rows, st);
}

// resident slots for the v2 kernel on the current device: 2 workgroups per CU (launch bounds), cached per device
inline int gemm_slots() {
    static int slots[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 512;
    if (slots[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        slots[dev] = 2 * n;
    }
    return slots[dev];
}

template <bool AK, bool BKc>
int launch(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc, float beta,
           int split_k, hipStream_t st) {
    const int tm = ds2_cdiv(M, BM), tn = ds2_cdiv(N, BN);
    if (split_k == 0) {  // auto: when M*N alone cannot fill the chip (256 CUs x 4 resident 256-thread workgroups), split K.
                         // Measured on the weight-gradient shapes (tools/attic/gemm_split_sweep.py): many small work items
                         // beat one round of big ones -- 4800x800x4050 TN: split 3 376 us, 8 318 us, 12 313 us -- so
                         // aim at ~3 items per slot while keeping >= 320 k (20 slabs) per item.
        // Round 2, deep-K shapes with up to two rounds of tiles (dX at B = 32: 13000 x 800 x 4800 = 714 tiles): 1 split
        // 1090 us, 4 959 us, 11 849 us -- ~8 items per slot, same floor per item.
        const int tiles = tm * tn;
        split_k = 1;
        if ((tiles < 512 && K >= 512) || (tiles < 2048 && K >= 2048)) {
            // (the split-operand kernels run a work item ~1.6x faster, so prologue, epilogue and atomics weigh more: stand-alone
            // dX 242 -> 213 us and dW_ih 242 -> 226 us with ~1024 items instead of ~8192, whole step -0.9 %)
            static const int env_target = ds2_tune_env("DS2_GEMM_SPLIT_TARGET") ? atoi(ds2_tune_env("DS2_GEMM_SPLIT_TARGET")) : 0;
            const int target = env_target > 0 ? env_target : (gemm_split_mode() != 0 ? 1024 : 8192);
            split_k = target / tiles;
            const int max_split = K / 320 > 1 ? K / 320 : 1;
            if (split_k > max_split) split_k = max_split;
            if (split_k > 32) split_k = 32;
        }
    }
    if (split_k < 1) split_k = 1;
    if (gemm_split_mode() != 0 && (lda % 4 == 0) && (ldb % 4 == 0) && (((uintptr_t)A & 15) == 0) && (((uintptr_t)B & 15) == 0) &&
        ((!AK && !BKc) || K % BK == 0)) {
        int kper = ds2_cdiv(ds2_cdiv(K, split_k), BK) * BK;
        if (kper < BK) kper = BK;
        const int nsplit = ds2_cdiv(K, kper);
        const unsigned long long abytes = 4ull * (AK ? (unsigned long long)(M - 1) * lda + K : (unsigned long long)(K - 1) * lda + M);
        const unsigned long long bbytes = 4ull * (BKc ? (unsigned long long)(N - 1) * ldb + K : (unsigned long long)(K - 1) * ldb + N);
        if (abytes >= 0x7FFFFFF0ull || bbytes >= 0x7FFFFFF0ull) return -1;
        // (round 5, built, measured and removed: the same loop on a 256 x 128 tile with EIGHT waves (4 x 2, 64 x 64 outputs each) and
    // one workgroup per CU -- global loads, split arithmetic and LDS stores per MFMA x 0.75, LDS operand reads per MFMA
    // unchanged; identical results.  us, 128 x 128 -> 256 x 128: 4096^3 NT / NN / TN 748 / 780 / 825 -> 714 / 740 / 780 (+5 %);
    // 47744 x 4800 x 800 (B = 64) 1953 -> 1903, 13000 x 4800 x 800 (B = 32) 561 -> 543 (+3 %); 4050 x 4800 x 800 (B = 10)
    // 179 -> 198 (-10 %: 608 tiles on 256 single-workgroup CUs, nothing fills the last round's bubbles).  3 % of the GEMMs at
    // B = 32 / 64 did not pay for a second kernel family; what is left in the loop is the LDS operand reads and the per-slab
    // barrier, i.e. a larger PER-WAVE tile: one wave per SIMD and software-pipelined fragment reads)
    // (round 5, measured and not kept: s_setprio 1 / 3 around a slab's MFMA block, 0 in front of the barrier -- of the two
    // workgroups a CU holds, the one that can feed the matrix pipe issues first.  Stand-alone, same box, us: dX (NN) 4240 x 800
    // x 4800 208 -> 201, dW_ih (TN) 218 -> 209, at 12500 rows 624 -> 607 / 582 -> 558, grouped dW_hh 198 -> 189, the input
    // projection (NT) and 4096^3 within 1 %; the whole step, three alternating pairs of 60 steps: B = 10 14.31 / 14.36 / 14.52
    // without against 14.87 / 14.51 / 14.27 ms with, B = 32 31.05 / 31.31 against 31.31 / 30.91, B = 64 x 15 s 99.3 against
    // 100.2 -- nothing: inside the step these GEMMs share the chip with other kernels, not with a second workgroup of their own)
    // (stream-K runs for a partial last round of tiles, as gemm_f32_v2_kernel has them, were built and measured: gi at
        // B = 10, 1292 tiles on 512 slots, 209 -> 204 us, 4096 x 4736 x 800 180 -> 189 us -- these kernels run against the
        // chip's power limit, idle slots give their share back as clock; not kept)
        // (round 6, built, measured and removed -- git history: gemm_bf16x_big_kernel -- the 256 x 128 tile with ONE wave per SIMD:
        // 2 x 2 waves of 128 x 64 outputs = 256 accumulator registers (hi + lo) in the AGPR half, 18 fragment reads for 48 MFMAs
        // per slab instead of 12 for 24, global loads / split / LDS stores per MFMA x 0.75, 86 KB of LDS, no spills; identical
        // results.  With the 128 x 128 kernel's issue-order hints: 4096^3 177 -> 167 TFLOP/s (-4 %); input projection at B = 10 /
        // 64: 194 -> 259 us / 1962 -> 2191 (-25 / -10 %); dX 222 -> 274 / 1985 -> 2518; dW_ih 237 -> 310 / 2302 -> 2877.  A single
        // wave per SIMD has nobody to fill its own waits (fragment reads behind every slab's barrier, the split's dependent
        // chains), and round 5's eight-wave form of the same tile -- two waves per SIMD -- gained 3-5 % at most: the loop is
        // within 1.5x of the 290 TFLOP/s the power-limited clock allows any bf16 loop, and the tile is not what is left)
        dim3 grid(tm * tn, nsplit), block(256);
        const int atomic = nsplit > 1 ? 1 : 0;
        if (atomic && beta == 0.f) zero_rows(C, ldc, N, M, st);
        if (gemm_split_mode() == 6)
            hipLaunchKernelGGL((gemm_bf16x_kernel<AK, BKc, 6>), grid, block, 0, st, M, N, K, A, lda, B, ldb, C, ldc, beta, tn,
                               kper, atomic, (unsigned int)abytes, (unsigned int)bbytes);
        else
            hipLaunchKernelGGL((gemm_bf16x_kernel<AK, BKc, 9>), grid, block, 0, st, M, N, K, A, lda, B, ldb, C, ldc, beta, tn,
                               kper, atomic, (unsigned int)abytes, (unsigned int)bbytes);
        return 0;
    }
    const bool vec0 = (lda % 4 == 0) && (ldb % 4 == 0) && (((uintptr_t)A & 15) == 0) && (((uintptr_t)B & 15) == 0) &&
                      ((!AK && !BKc) || K % 4 == 0);
    static const bool v2_on = !(ds2_tune_env("DS2_GEMM_V2") && ds2_tune_env("DS2_GEMM_V2")[0] == '0');    // A/B timing switch
    // (a 256 x 128 kernel with three LDS slab buffers for this family -- gemm_f32_v3_kernel, round 2: 134 against 124 TFLOP/s on
    // 4096^3, behind the stream-K kernel on the model's one-round shapes -- went when the split-operand family became the
    // default; removed in round 5)
    const int tiles_all = tm * tn, slots_all = gemm_slots();
    if (split_k == 1 && v2_on && tiles_all > slots_all && tiles_all % slots_all != 0 && ds2_cdiv(K, BK) >= 16 && beta == 0.f) {
        // whole-tile workgroups for as many complete rounds of the resident slots as there are, stream-K runs for the rest
        const unsigned long long abytes = 4ull * (AK ? (unsigned long long)(M - 1) * lda + K : (unsigned long long)(K - 1) * lda + M);
        const unsigned long long bbytes = 4ull * (BKc ? (unsigned long long)(N - 1) * ldb + K : (unsigned long long)(K - 1) * ldb + N);
        if (abytes >= 0x7FFFFFF0ull || bbytes >= 0x7FFFFFF0ull) return -1;
        const int tiles = tm * tn, slots = gemm_slots(), nslab = ds2_cdiv(K, BK);
        int dp = tiles, sk_wgs = 0, sk_units = 0, sk_total = 0;
        const int rem = tiles % slots;
        // (when the tiles fill whole rounds the two-buffer kernel below is ~4 % faster on 4096^3: its shorter loop body
        // wins once there is no partial round to balance)
        const bool hybrid = tiles > slots && rem != 0 && nslab >= 16 && beta == 0.f;
        if (hybrid) {
            dp = tiles - rem;
            sk_total = rem * nslab;
            sk_wgs = sk_total / 16 < slots ? sk_total / 16 : slots;         // at least 16 slabs per run
            if (sk_wgs < 1) sk_wgs = 1;
            sk_units = ds2_cdiv(sk_total, sk_wgs);
            if (sk_units > nslab) {                                          // a run may cross ONE tile boundary only
                sk_units = nslab;
            }
            sk_wgs = ds2_cdiv(sk_total, sk_units);
            const int r0 = (dp / tn) * BM;                                   // first row that holds a stream-K tile
            zero_rows(C + (size_t)r0 * ldc, ldc, N, M - r0, st);
        }
        dim3 grid(dp + sk_wgs), block(256);
        if (vec0)
            hipLaunchKernelGGL((gemm_f32_v2_kernel<AK, BKc, true>), grid, block, 0, st, M, N, K, A, lda, B, ldb, C, ldc,
                               beta, tn, dp, sk_units, sk_total, (unsigned int)abytes, (unsigned int)bbytes);
        else
            hipLaunchKernelGGL((gemm_f32_v2_kernel<AK, BKc, false>), grid, block, 0, st, M, N, K, A, lda, B, ldb, C, ldc,
                               beta, tn, dp, sk_units, sk_total, (unsigned int)abytes, (unsigned int)bbytes);
        return 0;
    }
    int kper = ds2_cdiv(ds2_cdiv(K, split_k), BK) * BK;
    if (kper < BK) kper = BK;
    const int nsplit = ds2_cdiv(K, kper);
    const bool vec = (lda % 4 == 0) && (ldb % 4 == 0) && (((uintptr_t)A & 15) == 0) && (((uintptr_t)B & 15) == 0) &&
                     ((!AK && !BKc) || K % 4 == 0);
    // bytes each descriptor must cover (rows x ld, last row only as wide as it is used)
    const unsigned long long abytes = 4ull * (AK ? (unsigned long long)(M - 1) * lda + K : (unsigned long long)(K - 1) * lda + M);
    const unsigned long long bbytes = 4ull * (BKc ? (unsigned long long)(N - 1) * ldb + K : (unsigned long long)(K - 1) * ldb + N);
    if (abytes >= 0x7FFFFFF0ull || bbytes >= 0x7FFFFFF0ull) return -1;
    dim3 grid(tm * tn, nsplit), block(256);
    const int atomic = nsplit > 1 ? 1 : 0;
    if (atomic && beta == 0.f)  // partial products are accumulated with atomics: start from zero
        zero_rows(C, ldc, N, M, st);
    if (vec)
        hipLaunchKernelGGL((gemm_f32_kernel<AK, BKc, true>), grid, block, 0, st, M, N, K, A, lda, B, ldb, C, ldc,
                           beta, tn, kper, atomic, (unsigned int)abytes, (unsigned int)bbytes);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<AK, BKc, false>), grid, block, 0, st, M, N, K, A, lda, B, ldb, C, ldc,
                           beta, tn, kper, atomic, (unsigned int)abytes, (unsigned int)bbytes);
    return 0;
}

}  // namespace

extern "C" int ds2_gemm_f32(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda,
                            const float* B, int ldb, float* C, int ldc, float beta, int split_k, void* stream) {
    DS2_CHECK_ARG(M > 0 && N > 0 && K > 0);
    DS2_CHECK_ARG(A && B && C);
    DS2_CHECK_ARG(beta == 0.f || beta == 1.f);
    DS2_CHECK_ARG(split_k >= 0);
    DS2_CHECK_ARG(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (!trans_a && !trans_b) rc = launch<true, false>(M, N, K, A, lda, B, ldb, C, ldc, beta, split_k, st);
    else if (!trans_a && trans_b) rc = launch<true, true>(M, N, K, A, lda, B, ldb, C, ldc, beta, split_k, st);
    else if (trans_a && !trans_b) rc = launch<false, false>(M, N, K, A, lda, B, ldb, C, ldc, beta, split_k, st);
    else rc = launch<false, true>(M, N, K, A, lda, B, ldb, C, ldc, beta, split_k, st);
    if (rc != 0) {
        ds2_set_error("ds2_gemm_f32: operand larger than 2 GB is not supported");
        return DS2_ERR_UNSUPPORTED;
    }
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" int ds2_gemm_split_mode(int mode) {
    if (mode == 0 || mode == 6 || mode == 9) gemm_split_mode_ref() = mode;
    return gemm_split_mode();
}

extern "C" int ds2_gemm_f32_tn_group(int count, const float* const* A_host, const int* lda_host, const int* M_host,
                                     const float* const* B_host, const int* ldb_host, float* const* C_host,
                                     const int* ldc_host, int N, int K, int accumulate, void* stream) {
    DS2_CHECK_ARG(count >= 1 && count <= 4 && A_host && lda_host && M_host && B_host && ldb_host && C_host && ldc_host);
    DS2_CHECK_ARG(N > 0 && K > 0);
    GemmGroup g;
    int max_tiles = 0, tot_tiles = 0;
    const int tn = ds2_cdiv(N, BN);
    bool vec = true;
    for (int p = 0; p < 4; ++p) {
        const int q = p < count ? p : 0;                        // unused slots repeat problem 0 (never launched)
        DS2_CHECK_ARG(A_host[q] && B_host[q] && C_host[q] && M_host[q] > 0 && lda_host[q] >= M_host[q] && ldb_host[q] >= N &&
                      ldc_host[q] >= N);
        g.A[p] = A_host[q];
        g.B[p] = B_host[q];
        g.C[p] = C_host[q];
        g.M[p] = M_host[q];
        g.lda[p] = lda_host[q];
        g.ldb[p] = ldb_host[q];
        g.ldc[p] = ldc_host[q];
        const unsigned long long ab = 4ull * ((unsigned long long)(K - 1) * lda_host[q] + M_host[q]);
        const unsigned long long bb = 4ull * ((unsigned long long)(K - 1) * ldb_host[q] + N);
        if (ab >= 0x7FFFFFF0ull || bb >= 0x7FFFFFF0ull) {
            ds2_set_error("ds2_gemm_f32_tn_group: operand larger than 2 GB is not supported");
            return DS2_ERR_UNSUPPORTED;
        }
        g.a_bytes[p] = (unsigned int)ab;
        g.b_bytes[p] = (unsigned int)bb;
        vec = vec && (lda_host[q] % 4 == 0) && (ldb_host[q] % 4 == 0) && (((uintptr_t)A_host[q] & 15) == 0) &&
              (((uintptr_t)B_host[q] & 15) == 0);
        if (p < count) {
            const int t = ds2_cdiv(M_host[q], BM) * tn;
            tot_tiles += t;
            if (t > max_tiles) max_tiles = t;
        }
    }
    // split K so that the launch has ~3 work items per resident slot (the rule of the single launch), >= 320 k per item
    int split_k = 1;
    if (tot_tiles < 512 && K >= 512) {
        static const int env_target = ds2_tune_env("DS2_GEMM_GROUP_TARGET") ? atoi(ds2_tune_env("DS2_GEMM_GROUP_TARGET")) : 0;
        const int target = env_target > 0 ? env_target : (gemm_split_mode() != 0 ? 1024 : 3072);
        split_k = target / tot_tiles;
        const int max_split = K / 320 > 1 ? K / 320 : 1;
        if (split_k > max_split) split_k = max_split;
        if (split_k > 32) split_k = 32;
        if (split_k < 1) split_k = 1;
    }
    int kper = ds2_cdiv(ds2_cdiv(K, split_k), BK) * BK;
    if (kper < BK) kper = BK;
    const int nsplit = ds2_cdiv(K, kper);
    hipStream_t st = (hipStream_t)stream;
    for (int p = 0; p < count && !accumulate; ++p)               // partial products are accumulated with atomics
        zero_rows(g.C[p], g.ldc[p], N, g.M[p], st);
    dim3 grid(max_tiles, nsplit, count), block(256);
    if (vec && gemm_split_mode() == 6)
        hipLaunchKernelGGL((gemm_bf16x_tn_group_kernel<6>), grid, block, 0, st, g, N, K, tn, kper);
    else if (vec && gemm_split_mode() == 9)
        hipLaunchKernelGGL((gemm_bf16x_tn_group_kernel<9>), grid, block, 0, st, g, N, K, tn, kper);
    else if (vec)
        hipLaunchKernelGGL((gemm_f32_tn_group_kernel<true>), grid, block, 0, st, g, N, K, tn, kper);
    else
        hipLaunchKernelGGL((gemm_f32_tn_group_kernel<false>), grid, block, 0, st, g, N, K, tn, kper);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

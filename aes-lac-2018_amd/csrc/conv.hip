// Direct (im2col-free) 2-D convolutions of the DeepSpeech2 front stack on the f32 MFMA, gfx950.
//
// Layout: NCHW with TIME innermost, so that the MFMA operand that walks output time steps is a
// contiguous (stride-1 or stride-2) window of an input row -- no im2col matrix ever exists.
//
//   forward   D[co][t]  += Wt[k][co] * in[b][ci][SF*d+kf][ST*t+kt-PADT]      k = (ci,kf,kt)
//   dgrad     D[ci][t1] += Wd[k][ci] * dout[b][co][(f-kf)/2][t1-kt]          k = (co,kf,kt), kf = f (mod 2)
//   wgrad     D[co][n]  += dout[b][co][d][t] * in[b][ci][SF*d+kf][ST*t+kt-PADT], n = (ci,kf,kt), k = (b,d,t)
//
// v_mfma_f32_32x32x2_f32: lanes 0-31 feed k even, lanes 32-63 k odd.  The 11 time taps are padded
// to 12 (tap 11 has zero weight) so that a k pair never straddles a filter row; the filter is
// re-laid-out once per call into [.. k ..][32] so the A fetch is one coalesced 128-B read per
// half-wave.  One wave = one 32 x (32*NT) output tile; 4 independent waves per workgroup walk
// neighbouring frequency rows so their input windows overlap in L1/L2.
#include <stdlib.h>

#include "split_bf16.h"

namespace {

constexpr int KTP = 12;  // padded time taps

// ---------------------------------------------------------------------------- filter re-layout
// mode 0 (forward):  wt[((ci*KF+kf)*12+kt)*32 + co] = w[((co*CIN+ci)*KF+kf)*KT+kt]
// mode 1 (dgrad):    wt[((co*KF+kf)*12+kt)*32 + ci] = w[((co*CIN+ci)*KF+kf)*KT+kt]   (CIN = 32)
__global__ __launch_bounds__(256) void conv_wt_layout_kernel(const float* __restrict__ w, int CIN, int KF, int KT,
                                                             int mode, float* __restrict__ wt) {
    const int total = (mode == 0 ? CIN : 32) * KF * KTP * 32;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int x = i & 31;
        int r = i >> 5;
        const int kt = r % KTP;
        r /= KTP;
        const int kf = r % KF;
        const int y = r / KF;
        const int co = mode == 0 ? x : y, ci = mode == 0 ? y : x;
        wt[i] = kt < KT ? w[(((size_t)co * CIN + ci) * KF + kf) * KT + kt] : 0.f;
    }
}

// ---------------------------------------------------------------------------- forward
// KS = waves that share one output tile, each taking 1/KS of the filter rows (K split inside the workgroup, partial
// tiles added through LDS): at B = 10 a layer has only 2.7 tiles per SIMD, too few waves to cover the operand loads' latency
// (matrix pipes 57 % busy) and a coarse last round; with KS = 2 twice as many waves of half the length are in flight.
template <int CIN, int KF, int SF, int ST, int PADT, int NT, int KS>
__global__ __launch_bounds__(256) void conv_fwd_kernel(const float* __restrict__ in, const float* __restrict__ wt,
                                                       const float* __restrict__ bias, int B, int FIN, int TIN,
                                                       int FOUT, int TOUT, int ttiles, float* __restrict__ out) {
    __shared__ float part[KS > 1 ? (4 / KS) * (KS - 1) * NT * 16 * 64 : 1];
    const int lane = threadIdx.x & 63;
    // readfirstlane: lets hipcc see that everything derived from the wave id is wave-uniform (SGPR buffer
    // descriptors instead of a waterfall loop per load -- cdna_hip_programming.md T20)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int slot = wave / KS, ks = wave % KS;          // tile slot of the workgroup, K share
    const long ntiles = (long)B * ttiles * FOUT;
    const long tile_raw = (long)blockIdx.x * (4 / KS) + slot;
    const bool tile_ok = tile_raw < ntiles;
    if (KS == 1 && !tile_ok) return;
    const long tile = tile_ok ? tile_raw : 0;
    const int d = (int)(tile % FOUT);
    const int tt = (int)((tile / FOUT) % ttiles);
    const int b = (int)(tile / ((long)FOUT * ttiles));
    const int t0 = tt * 32 * NT;

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // time index of this lane's output column for tap 0 (per N tile)
    int tbase[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) tbase[i] = ST * (t0 + 32 * i + lr) - PADT + lh;

    // rows r = (ci, kf) of the filter, walked with a one-row register prefetch: the 6 weight pairs and
    // 6*NT input pairs of row r+1 are in flight while the 6*NT MFMAs of row r issue.
    constexpr int NROWS_ALL = CIN * KF;
    const int r_lo = (NROWS_ALL * ks) / KS, NROWS = tile_ok ? (NROWS_ALL * (ks + 1)) / KS - r_lo : 0;
    const float* wp = wt + lh * 32 + lr;
    const float* inb = in + ((size_t)b * CIN * FIN + (size_t)SF * d) * TIN;
    float a_nxt[KTP / 2], v_nxt[NT][KTP / 2];
    auto fetch = [&](int rr) {
        const int r = r_lo + rr;
        const int ci = r / KF, kf = r - ci * KF;
        // the input row as a buffer resource: out-of-range time steps (the conv padding, the ragged tile end)
        // read back 0 from the hardware range check -- no compare, no select and, above all, no branch around
        // the load (hipcc turns a guarded or selected load into a branch + vmcnt(0) per element)
        const __amdgpu_buffer_rsrc_t row = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(inb + ((size_t)ci * FIN + kf) * TIN), 0, TIN * 4, 0x00020000);
        const float* w = wp + (size_t)r * (KTP * 32);
#pragma unroll
        for (int kp = 0; kp < KTP / 2; ++kp) {
            a_nxt[kp] = w[kp * 64];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                v_nxt[i][kp] = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(row, (tbase[i] + 2 * kp) * 4, 0, 0));
            }
        }
    };
    if (NROWS > 0) fetch(0);
    for (int r = 0; r < NROWS; ++r) {
        float a_cur[KTP / 2], v_cur[NT][KTP / 2];
#pragma unroll
        for (int kp = 0; kp < KTP / 2; ++kp) {
            a_cur[kp] = a_nxt[kp];
#pragma unroll
            for (int i = 0; i < NT; ++i) v_cur[i][kp] = v_nxt[i][kp];
        }
        if (r + 1 < NROWS) fetch(r + 1);
#pragma unroll
        for (int kp = 0; kp < KTP / 2; ++kp)
#pragma unroll
            for (int i = 0; i < NT; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[kp], v_cur[i][kp], acc[i], 0, 0, 0);
    }
    if constexpr (KS > 1) {                              // shares 1 .. KS-1 hand their partial tile to share 0 through LDS
        float* mine = part + ((size_t)(slot * (KS - 1) + (ks > 0 ? ks - 1 : 0)) * NT * 16) * 64 + lane;
        if (ks > 0) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[(i * 16 + r) * 64] = acc[i][r];
        }
        __syncthreads();
        if (ks > 0 || !tile_ok) return;
#pragma unroll
        for (int k2 = 0; k2 < KS - 1; ++k2) {
            const float* theirs = part + ((size_t)(slot * (KS - 1) + k2) * NT * 16) * 64 + lane;
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] += theirs[(i * 16 + r) * 64];
        }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int t = t0 + 32 * i + lr;
        if (t >= TOUT) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * lh;
            out[(((size_t)b * 32 + co) * FOUT + d) * TOUT + t] = acc[i][r] + bias[co];
        }
    }
}

// ---------------------------------------------------------------------------- conv2 dgrad
// d_in (B,32,FIN,T1) from d_out (B,32,FOUT,T); kernel (32,32,KF,11), stride (2,1), no padding
template <int KF, int NT, int KS>
__global__ __launch_bounds__(256) void conv2_dgrad_kernel(const float* __restrict__ dout, const float* __restrict__ wd,
                                                          int B, int FIN, int T1, int FOUT, int T, int ttiles,
                                                          float* __restrict__ din) {
    __shared__ float part[KS > 1 ? (4 / KS) * (KS - 1) * NT * 16 * 64 : 1];   // K split inside the workgroup: see conv_fwd_kernel
    const int lane = threadIdx.x & 63;
    // readfirstlane: lets hipcc see that everything derived from the wave id is wave-uniform (SGPR buffer
    // descriptors instead of a waterfall loop per load -- cdna_hip_programming.md T20)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int slot = wave / KS, ks = wave % KS;
    const long ntiles = (long)B * ttiles * FIN;
    const long tile_raw = (long)blockIdx.x * (4 / KS) + slot;
    const bool tile_ok = tile_raw < ntiles;
    if (KS == 1 && !tile_ok) return;
    const long tile = tile_ok ? tile_raw : 0;
    const int f = (int)(tile % FIN);
    const int tt = (int)((tile / FIN) % ttiles);
    const int b = (int)(tile / ((long)FIN * ttiles));
    const int t0 = tt * 32 * NT;

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    int tbase[NT];  // t1 - kt with kt = 2*kp + lh
#pragma unroll
    for (int i = 0; i < NT; ++i) tbase[i] = t0 + 32 * i + lr - lh;

    // valid filter rows for this input row f: kf = f (mod 2) with 0 <= (f-kf)/2 < FOUT
    const int kf_first = max(f & 1, f - 2 * (FOUT - 1));
    int nkf = 0;
    for (int kf = kf_first; kf < KF && kf <= f; kf += 2) ++nkf;
    const int nrows_all = 32 * nkf;
    const int r_lo = (nrows_all * ks) / KS, nrows = tile_ok ? (nrows_all * (ks + 1)) / KS - r_lo : 0;
    float a_nxt[KTP / 2], v_nxt[NT][KTP / 2];
    auto fetch = [&](int rr) {
        const int r = r_lo + rr;
        const int co = r / nkf, kf = kf_first + 2 * (r - co * nkf);
        const int d = (f - kf) >> 1;
        const __amdgpu_buffer_rsrc_t row = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(dout + (((size_t)b * 32 + co) * FOUT + d) * T), 0, T * 4, 0x00020000);
        const float* w = wd + ((size_t)(co * KF + kf) * KTP) * 32 + lh * 32 + lr;
#pragma unroll
        for (int kp = 0; kp < KTP / 2; ++kp) {
            a_nxt[kp] = w[kp * 64];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                v_nxt[i][kp] = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(row, (tbase[i] - 2 * kp) * 4, 0, 0));
            }
        }
    };
    if (nrows > 0) fetch(0);
    for (int r = 0; r < nrows; ++r) {
        float a_cur[KTP / 2], v_cur[NT][KTP / 2];
#pragma unroll
        for (int kp = 0; kp < KTP / 2; ++kp) {
            a_cur[kp] = a_nxt[kp];
#pragma unroll
            for (int i = 0; i < NT; ++i) v_cur[i][kp] = v_nxt[i][kp];
        }
        if (r + 1 < nrows) fetch(r + 1);
#pragma unroll
        for (int kp = 0; kp < KTP / 2; ++kp)
#pragma unroll
            for (int i = 0; i < NT; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[kp], v_cur[i][kp], acc[i], 0, 0, 0);
    }
    if constexpr (KS > 1) {
        float* mine = part + ((size_t)(slot * (KS - 1) + (ks > 0 ? ks - 1 : 0)) * NT * 16) * 64 + lane;
        if (ks > 0) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[(i * 16 + r) * 64] = acc[i][r];
        }
        __syncthreads();
        if (ks > 0 || !tile_ok) return;
#pragma unroll
        for (int k2 = 0; k2 < KS - 1; ++k2) {
            const float* theirs = part + ((size_t)(slot * (KS - 1) + k2) * NT * 16) * 64 + lane;
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] += theirs[(i * 16 + r) * 64];
        }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int t = t0 + 32 * i + lr;
        if (t >= T1) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = (r & 3) + 8 * (r >> 2) + 4 * lh;
            din[(((size_t)b * 32 + ci) * FIN + f) * T1 + t] = acc[i][r];
        }
    }
}

// ---------------------------------------------------------------------------- wgrad
// grid (ntiles_n, splits); one wave per workgroup slot: block = 256 threads = 4 waves, each wave
// takes every 4th (b,d) row of its split.  Results are atomically added into dW (zeroed by the host
// wrapper) -- float atomics make the last bits order-dependent.
template <int CIN, int KF, int KT, int SF, int ST, int PADT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ in, const float* __restrict__ dout,
                                                         int B, int FIN, int TIN, int FOUT, int TOUT, int nsplit,
                                                         float* __restrict__ dw, float* __restrict__ dbias) {
    constexpr int NTOT = CIN * KF * KT;
    const int lane = threadIdx.x & 63;
    // readfirstlane: lets hipcc see that everything derived from the wave id is wave-uniform (SGPR buffer
    // descriptors instead of a waterfall loop per load -- cdna_hip_programming.md T20)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int n = blockIdx.x * 32 + lr;
    const bool n_ok = n < NTOT;
    const int nn = n_ok ? n : 0;
    const int kt = nn % KT, kf = (nn / KT) % KF, ci = nn / (KT * KF);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;

    const int rows = B * FOUT;
    for (int row = blockIdx.y * 4 + wave; row < rows; row += nsplit * 4) {
        const int b = row / FOUT, d = row % FOUT;
        const float* ap = dout + (((size_t)b * 32 + lr) * FOUT + d) * TOUT;
        const float* bp = in + (((size_t)b * CIN + ci) * FIN + (size_t)SF * d + kf) * TIN;
        // 16 time steps per iteration: half-wave lh owns steps [t0 + 8 lh, t0 + 8 lh + 8) for A and B alike
        // (the k order inside the MFMA chain is free as long as both operands agree); the next group's
        // operands are fetched (two unaligned dwordx4 per operand) while the 8 MFMAs of this one issue.
        float a_nxt[8], v_nxt[8];
        auto fetch = [&](int t0) {
            const int t = t0 + 8 * lh;
            if (t + 7 < TOUT) {
                const f32x4u x0 = *reinterpret_cast<const f32x4u*>(ap + t);
                const f32x4u x1 = *reinterpret_cast<const f32x4u*>(ap + t + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a_nxt[e] = x0[e];
                    a_nxt[4 + e] = x1[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = ap[min(t + e, TOUT - 1)];
                    a_nxt[e] = (t + e < TOUT) ? x : 0.f;
                }
            }
            const int ti0 = ST * t + kt - PADT;
            if (ST == 1 && n_ok && t + 7 < TOUT && ti0 >= 0 && ti0 + 7 < TIN) {
                const f32x4u y0 = *reinterpret_cast<const f32x4u*>(bp + ti0);
                const f32x4u y1 = *reinterpret_cast<const f32x4u*>(bp + ti0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v_nxt[e] = y0[e];
                    v_nxt[4 + e] = y1[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ti = ti0 + ST * e;
                    const float x = bp[min(max(ti, 0), TIN - 1)];
                    v_nxt[e] = (n_ok && t + e < TOUT && ti >= 0 && ti < TIN) ? x : 0.f;
                }
            }
        };
        fetch(0);
        for (int t0 = 0; t0 < TOUT; t0 += 16) {
            float a[8], v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a[e] = a_nxt[e];
                v[e] = v_nxt[e];
            }
            if (t0 + 16 < TOUT) fetch(t0 + 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], v[e], acc, 0, 0, 0);
                bsum += a[e];
            }
        }
    }
    if (n_ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * lh;
            atomicAdd(&dw[(size_t)co * NTOT + n], acc[r]);
        }
    }
    if (blockIdx.x == 0) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (lh == 0) atomicAdd(&dbias[lr], bsum);
    }
}

// ---------------------------------------------------------------------------- conv2 wgrad, operands through LDS
// The direct kernel above feeds every MFMA from two per-lane global loads whose 64 lanes touch 32-64 different cache lines
// (lane = output channel / filter tap, 32 bytes of a row each): the L1's one-line-per-cycle tag path, not the matrix pipe,
// bounds it (75 TFLOP/s at B = 10), and the four waves of a workgroup read four different d(out) rows for the same taps.
// Here a workgroup owns 128 consecutive taps (one 32-tap tile per wave) and walks its (b, d) rows in chunks of 64 time
// steps: the d(out) chunk (32 channels x 64 steps, shared by the four waves) and the 13 input rows the 128 taps touch
// (each read at 11 time shifts) are loaded once, coalesced, into LDS (double-buffered, one barrier per chunk); the MFMA
// operands are conflict-free ds_read_b32.  Partial tiles are added into dW with float atomics as before.
// Round 4: the same kernel for conv1 (CIN = 1, 41 x 11 taps, time stride 2, 10 columns of zero padding on both sides: the
// input window of a 64-step chunk is 2 x 64 + 10 columns starting at column 2 t0 - 10, read by the MFMA operand at stride 2).
// conv1's direct kernel ran at 22 TFLOP/s (0.33 ms at the B = 10 bin mix, on the main stream behind everything else of the
// backward pass); its 451 taps are 4 groups of 128.
constexpr int WG_TC = 64, WG_PA = WG_TC + 1, WG_NROW = 13;   // pitch 65: bank = row + t
template <int CIN, int KF, int SF, int ST, int PADT>
__global__ __launch_bounds__(256) void conv_wgrad_lds_kernel(const float* __restrict__ in, const float* __restrict__ dout,
                                                             int B, int FIN, int TIN, int FOUT, int TOUT, int nsplit,
                                                             float* __restrict__ dw, float* __restrict__ dbias) {
    constexpr int KT = 11, NTOT = CIN * KF * KT;
    constexpr int WCOLS = ST * (WG_TC - 1) + KT;        // input columns a chunk touches per row: 74 (stride 1) / 137 (stride 2)
    constexpr int WG_PB = WCOLS | 1;                    // odd pitch (75 / 137): consecutive rows start on different banks
    constexpr int NB = WG_NROW * WCOLS, NBI = (NB + 255) / 256;
    __shared__ float sA[2][32 * WG_PA];
    __shared__ float sB[2][WG_NROW * WG_PB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.x * 128, r0 = n0 / KT;
    const int n = n0 + wave * 32 + lr;
    const bool n_ok = n < NTOT;
    const int nn = n_ok ? n : n0;
    const int boff = (nn / KT - r0) * WG_PB + nn % KT + ST * lh;   // this lane's tap inside the staged input rows (+ k parity)
    const int aoff = lr * WG_PA + lh;
    const int tchunks = (TOUT + WG_TC - 1) / WG_TC;
    const int rows = B * FOUT;
    const int myrows = (rows - (int)blockIdx.y + nsplit - 1) / nsplit;          // rows blockIdx.y, + nsplit, ...
    const int nchunks = myrows * tchunks;
    // staging roles: A -- channel tid / 8, eight steps from (tid % 8) * 8; B -- up to four scalars per thread
    const int a_co = tid >> 3, a_t = (tid & 7) * 8;
    // B staging: element idx = tid + 256 i of the [13 rows][WCOLS columns] window -> (row j, column x): fixed per thread for the launch
    int b_src[NBI], b_dst[NBI], b_x[NBI];               // source offset inside a (b, d) image (-1: nothing), LDS offset, column
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
        const int idx = tid + 256 * i;
        const int j = idx / WCOLS, x = idx - j * WCOLS;
        const int r = r0 + j, ci = r / KF, kf = r - ci * KF;
        b_src[i] = (idx < NB && r < CIN * KF) ? (ci * FIN + kf) * TIN + x : -1;
        b_dst[i] = idx < NB ? j * WG_PB + x : -1;
        b_x[i] = x;
    }
    float ra[8], rb[NBI];
    int g_row = blockIdx.y, g_tc = 0;                    // (row, time chunk) of the NEXT gload: chunks walk time fastest
    auto gload = [&]() {
        const int row = g_row, t0 = g_tc * WG_TC;
        if (++g_tc == tchunks) {
            g_tc = 0;
            g_row += nsplit;
        }
        const int b = row / FOUT, d = row % FOUT;
        const float* ap = dout + (((size_t)b * 32 + a_co) * FOUT + d) * TOUT + t0 + a_t;
        if (t0 + a_t + 7 < TOUT) {
            const f32x4u x0 = *reinterpret_cast<const f32x4u*>(ap), x1 = *reinterpret_cast<const f32x4u*>(ap + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ra[e] = x0[e];
                ra[4 + e] = x1[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) ra[e] = (t0 + a_t + e < TOUT) ? ap[e] : 0.f;
        }
        const int c0 = ST * t0 - PADT;                    // first input column of the window (negative: zero padding)
        const float* bp = in + ((size_t)b * CIN * FIN + (size_t)SF * d) * TIN + c0;
#pragma unroll
        for (int i = 0; i < NBI; ++i)
            rb[i] = (b_src[i] >= 0 && c0 + b_x[i] >= 0 && c0 + b_x[i] < TIN) ? bp[b_src[i]] : 0.f;
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sA[buf][a_co * WG_PA + a_t + e] = ra[e];
#pragma unroll
        for (int i = 0; i < NBI; ++i)
            if (b_dst[i] >= 0) sB[buf][b_dst[i]] = rb[i];
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
    if (nchunks > 0) {
        gload();
        lstore(0);
    }
    __syncthreads();
    const bool sum_bias = blockIdx.x == 0 && wave == 0;  // d(bias) = the d(out) row sums: one wave of the first tap group
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) gload();                    // in flight under this chunk's MFMAs
        const float* pa = sA[buf] + aoff;
        const float* pb = sB[buf] + boff;
        if (sum_bias) {
#pragma unroll 8
            for (int kp = 0; kp < WG_TC / 2; ++kp) {
                const float a = pa[2 * kp], v = pb[ST * 2 * kp];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v, acc, 0, 0, 0);
                bsum += a;
            }
        } else {
#pragma unroll 8
            for (int kp = 0; kp < WG_TC / 2; ++kp)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[2 * kp], pb[ST * 2 * kp], acc, 0, 0, 0);
        }
        if (c + 1 < nchunks) lstore(buf ^ 1);
        __syncthreads();
    }
    if (n_ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * lh;
            atomicAdd(&dw[(size_t)co * NTOT + n], acc[r]);
        }
    }
    if (sum_bias) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (lh == 0) atomicAdd(&dbias[lr], bsum);
    }
}

// ---------------------------------------------------------------------------- weight gradients on the bf16 matrix pipe
// The LDS form above is bound by the f32-input MFMA (32 v_mfma_f32_32x32x2 = 2048 pipe cycles per wave and 64-step chunk;
// 105-113 TFLOP/s).  Same workgroup geometry, staging and chunk walk here, but the products run as error-free split operands
// (split_bf16.h) on v_mfma_f32_32x32x16_bf16: a chunk is four 16-step instructions' worth x 6 products = 768 pipe cycles.
//   A = d(out), [32 channels][64 steps], shared by the four waves: split ONCE per chunk by the staging threads and kept in LDS
//       as bf16 planes, [channel][16-step slab][plane][16 steps], row pitch 400 B (the fragment reads, 16 bytes per lane, of
//       16 consecutive channels fall on 16 different 16-byte slots);
//   B = the input window, fp32 in LDS as before: lane = tap reads its 8 consecutive (stride ST) time steps at its own shift --
//       any alignment, so as eight ds_read_b32 -- and splits them in registers (44 vector instructions per 6 MFMAs: the
//       vector and the matrix pipe are about evenly loaded).
// d(bias) = the row sums of d(out): summed by the staging threads of the first tap group in fp32.
constexpr int WS_APITCH = 400;                      // bytes per channel row: 4 slabs x 3 planes x 32 B + 16
template <int CIN, int KF, int SF, int ST, int PADT>
__global__ __launch_bounds__(256) void conv_wgrad_split_kernel(const float* __restrict__ in, const float* __restrict__ dout,
                                                               int B, int FIN, int TIN, int FOUT, int TOUT, int nsplit,
                                                               float* __restrict__ dw, float* __restrict__ dbias) {
    constexpr int KT = 11, NTOT = CIN * KF * KT;
    constexpr int WCOLS = ST * (WG_TC - 1) + KT;
    constexpr int WG_PB = WCOLS | 1;
    constexpr int NB = WG_NROW * WCOLS, NBI = (NB + 255) / 256;
    __shared__ __attribute__((aligned(16))) char sA[2][32 * WS_APITCH];
    __shared__ float sB[2][WG_NROW * WG_PB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.x * 128, r0 = n0 / KT;
    const int n = n0 + wave * 32 + lr;
    const bool n_ok = n < NTOT;
    const int nn = n_ok ? n : n0;
    const int boff = (nn / KT - r0) * WG_PB + nn % KT + ST * 8 * lh;   // this lane's tap in the staged rows (+ its k half)
    const int aoff = lr * WS_APITCH + lh * 16;
    const int tchunks = (TOUT + WG_TC - 1) / WG_TC;
    const int rows = B * FOUT;
    const int myrows = (rows - (int)blockIdx.y + nsplit - 1) / nsplit;
    const int nchunks = myrows * tchunks;
    const int a_co = tid >> 3, a_t = (tid & 7) * 8;
    const int a_dst = a_co * WS_APITCH + (a_t >> 4) * 96 + ((a_t >> 3) & 1) * 16;
    int b_src[NBI], b_dst[NBI], b_x[NBI];
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
        const int idx = tid + 256 * i;
        const int j = idx / WCOLS, x = idx - j * WCOLS;
        const int r = r0 + j, ci = r / KF, kf = r - ci * KF;
        b_src[i] = (idx < NB && r < CIN * KF) ? (ci * FIN + kf) * TIN + x : -1;
        b_dst[i] = idx < NB ? j * WG_PB + x : -1;
        b_x[i] = x;
    }
    float ra[8], rb[NBI];
    int g_row = blockIdx.y, g_tc = 0;
    auto gload = [&]() {
        const int row = g_row, t0 = g_tc * WG_TC;
        if (++g_tc == tchunks) {
            g_tc = 0;
            g_row += nsplit;
        }
        const int b = row / FOUT, d = row % FOUT;
        const float* ap = dout + (((size_t)b * 32 + a_co) * FOUT + d) * TOUT + t0 + a_t;
        if (t0 + a_t + 7 < TOUT) {
            const f32x4u x0 = *reinterpret_cast<const f32x4u*>(ap), x1 = *reinterpret_cast<const f32x4u*>(ap + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ra[e] = x0[e];
                ra[4 + e] = x1[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) ra[e] = (t0 + a_t + e < TOUT) ? ap[e] : 0.f;
        }
        const int c0 = ST * t0 - PADT;
        const float* bp = in + ((size_t)b * CIN * FIN + (size_t)SF * d) * TIN + c0;
#pragma unroll
        for (int i = 0; i < NBI; ++i)
            rb[i] = (b_src[i] >= 0 && c0 + b_x[i] >= 0 && c0 + b_x[i] < TIN) ? bp[b_src[i]] : 0.f;
    };
    const bool sum_bias = blockIdx.x == 0;               // d(bias): the staging threads of the first tap group
    float bsum = 0.f;
    auto lstore = [&](int buf) {
        unsigned int pl[3][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) split3(ra[2 * c], ra[2 * c + 1], pl[0][c], pl[1][c], pl[2][c]);
#pragma unroll
        for (int q = 0; q < 3; ++q)
            *reinterpret_cast<uint4*>(sA[buf] + a_dst + q * 32) = make_uint4(pl[q][0], pl[q][1], pl[q][2], pl[q][3]);
        if (sum_bias) bsum += ((ra[0] + ra[1]) + (ra[2] + ra[3])) + ((ra[4] + ra[5]) + (ra[6] + ra[7]));
#pragma unroll
        for (int i = 0; i < NBI; ++i)
            if (b_dst[i] >= 0) sB[buf][b_dst[i]] = rb[i];
    };
    f32x16 acc, acc_lo;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = acc_lo[r] = 0.f;
    if (nchunks > 0) {
        gload();
        lstore(0);
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
#ifndef DS2_WGRAD_ABL
#define DS2_WGRAD_ABL 0        // timing-only ablations (results WRONG): 1 no MFMAs, 2 no operand split, 4 no global loads in the
#endif                         // loop, 8 no atomics, 16 no LDS stores in the loop
        if (c + 1 < nchunks && !(DS2_WGRAD_ABL & 4)) gload();                    // in flight under this chunk's MFMAs
        const char* pa = sA[buf] + aoff;
        const float* pb = sB[buf] + boff;
#pragma unroll
        for (int ks = 0; ks < WG_TC / 16; ++ks) {
            bf16x8 a[3], b[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) a[q] = *reinterpret_cast<const bf16x8*>(pa + ks * 96 + q * 32);
            float bv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bv[e] = pb[ST * (16 * ks + e)];
            unsigned int p[3][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (DS2_WGRAD_ABL & 2) {
                    p[0][e] = __builtin_bit_cast(unsigned int, bv[2 * e]);
                    p[1][e] = __builtin_bit_cast(unsigned int, bv[2 * e + 1]);
                    p[2][e] = p[0][e] ^ p[1][e];
                } else
                    split3(bv[2 * e], bv[2 * e + 1], p[0][e], p[1][e], p[2][e]);
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) b[q] = __builtin_bit_cast(bf16x8, make_uint4(p[q][0], p[q][1], p[q][2], p[q][3]));
            if (DS2_WGRAD_ABL & 1) {
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r + 4 * q] += __builtin_bit_cast(f32x4, a[q])[r] + __builtin_bit_cast(f32x4, b[q])[r];
            } else
                split_mfma2<6>(a, b, acc, acc_lo);
        }
        if (c + 1 < nchunks && !(DS2_WGRAD_ABL & 16)) lstore(buf ^ 1);
        __syncthreads();
    }
    if (n_ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (DS2_WGRAD_ABL & 8) {
                if (acc[r] + acc_lo[r] == 12345.678f) dw[(size_t)co * NTOT + n] = acc[r];
            } else
            atomicAdd(&dw[(size_t)co * NTOT + n], acc[r] + acc_lo[r]);
        }
    }
    if (sum_bias) {                                      // eight staging threads (adjacent lanes) per channel
        bsum += __shfl_xor(bsum, 1, 64);
        bsum += __shfl_xor(bsum, 2, 64);
        bsum += __shfl_xor(bsum, 4, 64);
        if ((tid & 7) == 0) atomicAdd(&dbias[a_co], bsum);
    }
}

struct ConvGeom {
    int cin, kf, kt, sf, st, padt, fin, fout;
};
inline ConvGeom geom(int which) {
    if (which == 1) return {1, 41, 11, 2, 2, 10, 161, 61};
    return {32, 21, 11, 2, 1, 0, 61, 21};
}
inline int conv_tout(int which, int tin) { return which == 1 ? (tin + 20 - 11) / 2 + 1 : tin - 10; }

}  // namespace

extern "C" size_t ds2_conv_wt_ws_floats(int which) {
    const ConvGeom g = geom(which);
    const size_t direct = (size_t)32 * g.kf * KTP * 32 > (size_t)g.cin * g.kf * KTP * 32 ? (size_t)32 * g.kf * KTP * 32
                                                                                       : (size_t)g.cin * g.kf * KTP * 32;
    const size_t split = which == 2 ? ds2_conv2_split_ws_floats() : 0;           // conv_split.hip's tables and filter image
    return direct > split ? direct : split;
}

extern "C" int ds2_conv_fwd(int which, const float* in, const float* weight, const float* bias, int B,
                            int t_in_frames, float* out, float* wt_ws, void* stream) {
    DS2_CHECK_ARG(which == 1 || which == 2);
    DS2_CHECK_ARG(in && weight && bias && out && wt_ws && B > 0);
    const ConvGeom g = geom(which);
    const int tin = t_in_frames, tout = conv_tout(which, tin);
    DS2_CHECK_ARG(tout > 0);
    hipStream_t st = (hipStream_t)stream;
    if (which == 2 && ds2_conv2_fwd_split(in, weight, bias, B, tin, out, wt_ws, st) == 0) {   // gather-GEMM, conv_split.hip
        DS2_CHECK_LAUNCH();
        return DS2_OK;
    }
    const int total = g.cin * g.kf * KTP * 32;
    hipLaunchKernelGGL(conv_wt_layout_kernel, dim3(ds2_cdiv(total, 256)), dim3(256), 0, st, weight, g.cin, g.kf, g.kt,
                       0, wt_ws);
    // tile width: 64 time steps per wave halves the filter traffic, 32 gives twice the waves; pick the one that
    // leaves fewer idle SIMD slots in the last round (1024 SIMDs)
    auto rounds = [&](int nt) {
        const long waves = (long)B * ds2_cdiv(tout, 32 * nt) * g.fout;
        return (double)((waves + 1023) / 1024) * nt;
    };
    const bool narrow = ds2_tune_env("DS2_CONV_NT") ? atoi(ds2_tune_env("DS2_CONV_NT")) == 1 : rounds(1) < rounds(2);
    // conv2: 64 time steps per wave always (half the filter traffic per output), the K split below supplies the waves
    const int nt = (which == 2 && !ds2_tune_env("DS2_CONV_NT")) ? 2 : (narrow ? 1 : 2);
    const int ttiles = ds2_cdiv(tout, 32 * nt);
    const long ntiles = (long)B * ttiles * g.fout;
    // K split inside the workgroup (two waves per tile) while the tiles alone give fewer than ~4 waves per SIMD: more waves
    // in flight to cover the operand loads, and a finer last round.  DS2_CONV_KS = 1 / 2 forces one (A/B timing, tests).
    // (measured, B = 10, conv2 forward, ms: T_in = 300 0.39 -> 0.27 (KS 4), 500 0.41 -> 0.35 (KS 2), 830 0.56 -> 0.51 (KS 2),
    // 1501 0.85 -> 0.81 (KS 2); conv1's 41-row tiles are too short to split)
    int ks = which == 1 ? 1 : (ntiles < 700 ? 4 : (ntiles < 4096 ? 2 : 1));
    if (ds2_tune_env("DS2_CONV_KS")) ks = atoi(ds2_tune_env("DS2_CONV_KS")) == 4 ? 4 : (atoi(ds2_tune_env("DS2_CONV_KS")) == 2 ? 2 : 1);
    dim3 grid((unsigned)((ntiles * ks + 3) / 4)), block(256);
#define DS2_CONV_FWD_GO(C, KF_, SF_, ST_, P_, N_)                                                                       \
    do {                                                                                                                \
        if (ks == 4)                                                                                                    \
            hipLaunchKernelGGL((conv_fwd_kernel<C, KF_, SF_, ST_, P_, N_, 4>), grid, block, 0, st, in, wt_ws, bias, B, g.fin, \
                               tin, g.fout, tout, ttiles, out);                                                         \
        else if (ks == 2)                                                                                               \
            hipLaunchKernelGGL((conv_fwd_kernel<C, KF_, SF_, ST_, P_, N_, 2>), grid, block, 0, st, in, wt_ws, bias, B, g.fin, \
                               tin, g.fout, tout, ttiles, out);                                                         \
        else                                                                                                            \
            hipLaunchKernelGGL((conv_fwd_kernel<C, KF_, SF_, ST_, P_, N_, 1>), grid, block, 0, st, in, wt_ws, bias, B, g.fin, \
                               tin, g.fout, tout, ttiles, out);                                                         \
    } while (0)
    if (which == 1 && nt == 2) DS2_CONV_FWD_GO(1, 41, 2, 2, 10, 2);
    else if (which == 1) DS2_CONV_FWD_GO(1, 41, 2, 2, 10, 1);
    // (a filter-through-LDS form of the direct kernel -- eight waves sharing an input channel's filter slice -- was 12 % faster
    // from B = 32 up, round 2; the gather-GEMM on the bf16 pipe, the default since round 3, beats both: removed in round 5)
    else if (nt == 2) DS2_CONV_FWD_GO(32, 21, 2, 1, 0, 2);
    else DS2_CONV_FWD_GO(32, 21, 2, 1, 0, 1);
#undef DS2_CONV_FWD_GO
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" size_t ds2_conv2_dgrad_ws_floats(int B, int T1) {
    const size_t direct = ds2_conv_wt_ws_floats(2), split = ds2_conv2_dgrad_split_ws_floats(B, T1);
    return direct > split ? direct : split;
}

extern "C" int ds2_conv2_dgrad(const float* d_out, const float* weight, int B, int T1, float* d_in, float* wt_ws,
                               size_t ws_floats, void* stream) {
    DS2_CHECK_ARG(d_out && weight && d_in && wt_ws && B > 0 && T1 > 10);
    DS2_CHECK_ARG(ws_floats >= ds2_conv_wt_ws_floats(2));
    hipStream_t st = (hipStream_t)stream;
    // the gather form needs room for its zero-bordered copy of d_out: an undersized workspace selects the direct kernel
    if (ws_floats >= ds2_conv2_dgrad_split_ws_floats(B, T1) &&
        ds2_conv2_dgrad_split(d_out, weight, B, T1, d_in, wt_ws, st) == 0) {      // gather-GEMM, conv_split.hip
        DS2_CHECK_LAUNCH();
        return DS2_OK;
    }
    const int T = T1 - 10;
    const int total = 32 * 21 * KTP * 32;
    hipLaunchKernelGGL(conv_wt_layout_kernel, dim3(ds2_cdiv(total, 256)), dim3(256), 0, st, weight, 32, 21, 11, 1,
                       wt_ws);
    auto rounds = [&](int nt) {
        const long waves = (long)B * ds2_cdiv(T1, 32 * nt) * 61;
        return (double)((waves + 1023) / 1024) * nt;
    };
    const bool narrow = ds2_tune_env("DS2_CONV_NT") ? atoi(ds2_tune_env("DS2_CONV_NT")) == 1 : rounds(1) < rounds(2);
    const int nt = ds2_tune_env("DS2_CONV_NT") ? (narrow ? 1 : 2) : 2;
    const int ttiles = ds2_cdiv(T1, 32 * nt);
    const long ntiles = (long)B * ttiles * 61;
    // K split inside the workgroup: as in ds2_conv_fwd.  Measured, B = 10, ms, 32-step tiles unsplit -> 64-step tiles with
    // four K shares: T_in = 300 0.28 -> 0.25, 830 0.64 -> 0.48, 1100 0.84 -> 0.61, 1501 1.05 -> 0.77
    // (B = 32 / 64 at T_in = 1000: 2.09 -> 1.56 / 3.93 -> 3.06)
    int ks = 4;
    if (ds2_tune_env("DS2_CONV_KS")) ks = atoi(ds2_tune_env("DS2_CONV_KS")) == 4 ? 4 : (atoi(ds2_tune_env("DS2_CONV_KS")) == 2 ? 2 : 1);
    const dim3 grid((unsigned)((ntiles * ks + 3) / 4));
#define DS2_DGRAD_GO(N_, K_)                                                                                             \
    hipLaunchKernelGGL((conv2_dgrad_kernel<21, N_, K_>), grid, dim3(256), 0, st, d_out, wt_ws, B, 61, T1, 21, T, ttiles, d_in)
    if (nt == 2 && ks == 4) DS2_DGRAD_GO(2, 4);
    else if (nt == 2 && ks == 2) DS2_DGRAD_GO(2, 2);
    else if (nt == 2) DS2_DGRAD_GO(2, 1);
    else if (ks == 4) DS2_DGRAD_GO(1, 4);
    else if (ks == 2) DS2_DGRAD_GO(1, 2);
    else DS2_DGRAD_GO(1, 1);
#undef DS2_DGRAD_GO
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

extern "C" int ds2_conv_wgrad(int which, const float* in, const float* d_out, int B, int t_in_frames, float* d_weight,
                              float* d_bias, void* stream) {
    DS2_CHECK_ARG(which == 1 || which == 2);
    DS2_CHECK_ARG(in && d_out && d_weight && d_bias && B > 0);
    const ConvGeom g = geom(which);
    const int tin = t_in_frames, tout = conv_tout(which, tin);
    DS2_CHECK_ARG(tout > 0);
    hipStream_t st = (hipStream_t)stream;
    const int ntot = g.cin * g.kf * g.kt;
    DS2_HIP(hipMemsetAsync(d_weight, 0, (size_t)32 * ntot * sizeof(float), st));
    DS2_HIP(hipMemsetAsync(d_bias, 0, 32 * sizeof(float), st));
    const int ntn = ds2_cdiv(ntot, 32);
    const int rows = B * g.fout;
    int nsplit = ds2_cdiv(2048, ntn);
    if (nsplit > ds2_cdiv(rows, 4)) nsplit = ds2_cdiv(rows, 4);
    if (nsplit < 1) nsplit = 1;
    dim3 grid(ntn, nsplit), block(256);
    const bool lds_form = !(getenv("DS2_CONV_WGRAD_LDS") && getenv("DS2_CONV_WGRAD_LDS")[0] == '0');
    // DS2_CONV_WGRAD_BF16 = 0: the LDS form on the f32-input MFMA (A/B timing, tests); default: split operands on the bf16 pipe
    const bool split_form = lds_form && !(getenv("DS2_CONV_WGRAD_BF16") && getenv("DS2_CONV_WGRAD_BF16")[0] == '0');
    if (split_form) {
        int split = which == 1 ? 256 : 105;              // the LDS form's workgroup counts (below)
        const char* e = ds2_tune_env(which == 1 ? "DS2_CONV1_WGRAD_SPLIT" : "DS2_CONV_WGRAD_SPLIT");
        if (e) split = atoi(e);
        if (split > rows) split = rows;
        if (split < 1) split = 1;
        if (which == 1)
            hipLaunchKernelGGL((conv_wgrad_split_kernel<1, 41, 2, 2, 10>), dim3(ds2_cdiv(ntot, 128), split), block, 0, st, in,
                               d_out, B, g.fin, tin, g.fout, tout, split, d_weight, d_bias);
        else
            hipLaunchKernelGGL((conv_wgrad_split_kernel<32, 21, 2, 1, 0>), dim3(ds2_cdiv(ntot, 128), split), block, 0, st, in,
                               d_out, B, g.fin, tin, g.fout, tout, split, d_weight, d_bias);
    } else if (which == 1 && lds_form) {
        // conv1 through LDS (round 4): 4 groups of 128 taps x up to 256 row splits ~ 1000 workgroups, ~4 per CU
        int split = 256;
        if (ds2_tune_env("DS2_CONV1_WGRAD_SPLIT")) split = atoi(ds2_tune_env("DS2_CONV1_WGRAD_SPLIT"));
        if (split > rows) split = rows;
        if (split < 1) split = 1;
        hipLaunchKernelGGL((conv_wgrad_lds_kernel<1, 41, 2, 2, 10>), dim3(ds2_cdiv(ntot, 128), split), block, 0, st, in, d_out, B,
                           g.fin, tin, g.fout, tout, split, d_weight, d_bias);
    } else if (which == 1)
        hipLaunchKernelGGL((conv_wgrad_kernel<1, 41, 11, 2, 2, 10>), grid, block, 0, st, in, d_out, B, g.fin, tin,
                           g.fout, tout, nsplit, d_weight, d_bias);
    else if (lds_form) {
        // operands through LDS: 58 groups of 128 taps x 105 row splits = ~6000 workgroups, four rounds of the six a CU holds
        // (measured, ms, B = 10 / 32 / 64 / 8 x 15 s: 0.41 / 1.44 / 2.84 / 0.56; the direct kernel: 0.57 / 2.15 / 4.35 / 0.79;
        // 13 ... 53 splits: 0.42 - 0.49 at B = 10).  DS2_CONV_WGRAD_LDS = 0: the direct kernel, for A/B timing and tests
        const int ngroups = ds2_cdiv(ntot, 128);
        int split = 105;
        if (ds2_tune_env("DS2_CONV_WGRAD_SPLIT")) split = atoi(ds2_tune_env("DS2_CONV_WGRAD_SPLIT"));
        if (split > rows) split = rows;
        if (split < 1) split = 1;
        hipLaunchKernelGGL((conv_wgrad_lds_kernel<32, 21, 2, 1, 0>), dim3(ngroups, split), block, 0, st, in, d_out, B, g.fin, tin,
                           g.fout, tout, split, d_weight, d_bias);
    }
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<32, 21, 11, 2, 1, 0>), grid, block, 0, st, in, d_out, B, g.fin, tin,
                           g.fout, tout, nsplit, d_weight, d_bias);
    DS2_CHECK_LAUNCH();
    return DS2_OK;
}

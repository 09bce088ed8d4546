// conv2 forward and data gradient as gather-GEMMs on the bf16 matrix pipe (fp32 in / fp32 accumulate / fp32 out by
// error-free operand splitting, split_bf16.h), gfx950.
//
//   C[p, n] = sum_k A[base(p) + off(k)] * W[n][k],   n < 32
//
// Forward:  p = (b, fo, t) output position, n = output channel, k = (ci, kh, kw); A = conv2's input,
//           base = ((b 32) 61 + 2 fo) T1 + t, off = (ci 61 + kh) T1 + kw.
// Dgrad:    p = (b, f', t1) input position of one row parity (f = 2 f' + par), n = input channel, k = (oc, kh', kw) over the
//           filter rows of that parity (kh = 2 kh' + par); A = d(out) in a zero-bordered copy (10 columns and 10 rows on
//           every side: every tap of every position reads inside its utterance or a zero).
//
// No LDS and no barrier in the main loop.  v_mfma_f32_32x32x16_bf16 wants, per lane, 8 consecutive k of one row (row =
// lane & 31, k group = lane >> 5).  The 16 k of a step are ordered so that the second group is the first one moved by a
// CONSTANT distance D in memory -- the same 8 filter taps of the NEXT channel (D = one channel plane) -- so a lane's byte
// offset is fixed for the whole tile (position + (lane >> 5) D) and the 8 taps' offsets are wave-uniform: they ride in the
// buffer instruction's scalar offset, read from a small table.  The 8 taps of a group are two runs of 4 consecutive
// time taps (address = t + kw: four taps of one position are 16 contiguous bytes), so a lane gathers its 8 values with
// TWO 16-byte loads (4-byte aligned) -- the texture-address unit takes 16 cycles per wave-instruction whatever its width,
// and eight 4-byte gathers per row tile made that unit the bound (0.40 ms, no better than the direct kernel).  The lane
// splits the 8 values into the three bf16 terms in registers -- which IS the MFMA fragment -- and multiplies.  11 taps per
// filter row: taps 0-7 of a row are one group, taps 8-10 of two rows share a group (2 slots of 8 multiply zeros).  The filter is split and laid out in fragment order once per call by a small kernel
// (3 KB per step, the same for every wave: L1 / L2 hits, no vector work).  A wave owns 64 positions x 32 channels (two row
// tiles sharing the filter fragments, 12 MFMAs per step); the taps of one channel (231 forward; 121 / 110 per parity
// backward) are padded to a multiple of 8 with zero filter entries.
// The direct kernels of conv.hip feed the f32-input MFMA from the same 4-byte gathers, one MFMA per two k: 83 - 90 TFLOP/s.
#include <stdlib.h>

#include <type_traits>

#include "split_bf16.h"

namespace {

constexpr int G_OOB = 0x7FFFFFF0;
constexpr int SPARE = 4;                 // zero steps behind the last one (the pipeline reads NSET - 1 steps ahead)

struct GatherGeom {
    int M;                                   // positions
    int ngroups, npairs;                     // tap groups of 8 per channel, channel pairs (K = npairs x ngroups x 16)
    int pair_stride, D;                      // floats: two channel planes, one channel plane
    int row_len, rows_per_b;
    int a_base, a_b_stride, a_row_stride;
    int o_base, o_b_stride, o_row_stride, o_col_stride;
    int frows;                               // > 0 (data gradient, K not split over workgroups): filter rows of this parity; a
                                             // workgroup multiplies only the tap groups whose filter rows reach an output row
                                             // from one of ITS input rows
};

template <int NPROD, int NT>
__global__ __launch_bounds__(256, 2) void conv2_gather_kernel(const float* __restrict__ A, unsigned int a_bytes,
                                                              const int* __restrict__ tap_off,
                                                              const unsigned int* __restrict__ Wp, unsigned int w_bytes,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              GatherGeom g, int steps_per_split, int use_atomic,
                                                              int tiles0, int tiles1, const int* __restrict__ tap_off1,
                                                              const unsigned int* __restrict__ Wp1, unsigned int w_bytes1,
                                                              GatherGeom g1) {
    __shared__ int sh_o[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs; give every XCD a contiguous run of positions (its
    // L2 then holds the 21 input rows its neighbouring tiles share)
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int n0 = (tiles0 >> 3) + (xcd < (tiles0 & 7) ? 1 : 0);
    int tile = xcd * (tiles0 >> 3) + min(xcd, tiles0 & 7) + jx;
    // Two problems in one launch (the data gradient's two input-row parities, tiles0 and tiles1 tiles of positions: one tail
    // instead of two, and with the row-dependent step counts below twice the workgroups to balance over the chip's 512 slots).
    // An XCD takes ITS eighth of the first problem's positions, then its eighth of the second's -- the two read the same
    // rows of A (dealing the concatenated tiles out in eighths put a quarter of A through every L2: T_in = 400 0.211 -> 0.235 ms).
    if (jx >= n0) {
        const int n1 = (tiles1 >> 3) + (xcd < (tiles1 & 7) ? 1 : 0);
        if (jx - n0 >= n1) return;                                       // (the grid is padded to 8 x the fullest XCD's share)
        tile = xcd * (tiles1 >> 3) + min(xcd, tiles1 & 7) + jx - n0;
        g = g1;
        tap_off = tap_off1;
        Wp = Wp1;
        w_bytes = w_bytes1;
    }
    constexpr int WG_M = 128 * NT;                                       // positions per workgroup: 4 waves x NT row tiles of 32
    const int m0 = tile * WG_M;
    // Data gradient: input row f' of a parity takes filter row kh' through output row f' - kh', which exists for
    // 0 <= f' - kh' <= 20 -- the 10 top and 10 bottom input rows of a parity use 1 .. 10 of its 11 / 10 filter rows, every other
    // tap multiplies the zero border (31 % of the steps of a launch).  A workgroup's positions lie in a few adjacent rows (one or
    // two at the bench shapes): it walks the tap groups [glo, ghi) of the filter-row PAIRS those rows need (three groups per
    // pair, see tap_slot).
    int glo = 0, ghi = g.ngroups;
    if (g.frows > 0) {
        const int p0 = min(m0, g.M - 1), p1 = min(m0 + WG_M, g.M) - 1;
        const int ri0 = p0 / g.row_len, ri1 = p1 / g.row_len;
        const int r0 = ri0 % g.rows_per_b, r1 = ri1 % g.rows_per_b;
        if (ri0 / g.rows_per_b == ri1 / g.rows_per_b) {   // rows r0 .. r1 of ONE batch element (across two: the whole range)
            const int lo = max(0, r0 - 20), hi = min(g.frows - 1, r1);
            glo = 3 * (lo >> 1);
            ghi = min(g.ngroups, 3 * (hi >> 1) + 3);
        }
    }
    const int ngr = ghi - glo;
    const int nstep_all = g.npairs * ngr;                 // steps of this workgroup
    const int sbeg = g.frows > 0 ? 0 : blockIdx.y * steps_per_split;
    const int send = g.frows > 0 ? nstep_all : min(nstep_all, sbeg + steps_per_split);
    auto decode = [&](int p, int& va, int& vo) {
        va = G_OOB;
        vo = -1;
        if (p < g.M) {
            const int rowi = p / g.row_len, t = p - rowi * g.row_len;
            const int b = rowi / g.rows_per_b, r = rowi - b * g.rows_per_b;
            va = (g.a_base + b * g.a_b_stride + r * g.a_row_stride + t + lh * g.D) * 4;
            vo = g.o_base + b * g.o_b_stride + r * g.o_row_stride + t;
        }
    };
    int va[NT], vo_unused;
    if (tid < WG_M) {
        int a_, o_;
        decode(m0 + tid, a_, o_);
        sh_o[tid] = o_;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) decode(m0 + wave * 32 * NT + 32 * i + lr, va[i], vo_unused);
    __syncthreads();
    const int vw = (lr * 16 + lh * 8) * 2;                               // filter fragment: [step][plane][oc][k half][8] bf16
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int*>(Wp), 0, w_bytes, 0x00020000);

    f32x16 acc[NT], acc_lo[NT];                          // leading products / products with a residual term (split_mfma2)
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = acc_lo[i][r] = 0.f;

    // Software pipeline, NSET register sets: at the top of step s the loads of step s + NSET - 1 are issued (their scalar
    // offsets were read from the tap table a step earlier), then step s -- loaded NSET - 1 steps ago -- is split and
    // multiplied.  (The input does not fit one XCD's L2 next to its neighbours' share: a third of the L2 requests go on to
    // the Infinity Cache, and with one set in flight the waves spent half their cycles waiting for them.)
    constexpr int NSET = 4;
    int cp = sbeg / ngr, gq = glo + sbeg - cp * ngr;                     // (channel pair, tap group) of the next table read
    f32x4 ra[NSET][NT][2];                                               // [set][row tile][run of 4 taps]
    uint4 rw[NSET][3];
    int so[3];                                                           // two run offsets and the step's place in the filter image
    auto read_tab = [&](int (&o)[3]) {
        const int* tab = tap_off + gq * 2;
        const int pbase = cp * g.pair_stride;
        o[0] = (pbase + tab[0]) * 4;                                     // wave-uniform: scalar loads
        o[1] = (pbase + tab[1]) * 4;
        o[2] = (cp * g.ngroups + gq) * 3 * 1024;
        if (++gq == ghi) {
            gq = glo;
            ++cp;
        }
    };
    auto issue = [&](f32x4 (&a)[NT][2], uint4 (&w)[3], const int (&o)[3]) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                a[i][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsa, va[i], o[h], 0));
#pragma unroll
        for (int q = 0; q < 3; ++q)
            w[q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsw, vw, o[2] + q * 1024, 0));
    };
    // prologue: steps sbeg .. sbeg + NSET - 2 in flight, the table entry of step sbeg + NSET - 1 read
    read_tab(so);
    issue(ra[0], rw[0], so);
    read_tab(so);
    issue(ra[1], rw[1], so);
    read_tab(so);
    issue(ra[2], rw[2], so);
    read_tab(so);                                                        // for step sbeg + 3, issued by the loop's first step
    auto step = [&](int, auto PAR) {
        constexpr int cur = decltype(PAR)::value;                        // (s - sbeg) mod NSET
        constexpr int nxt = (cur + NSET - 1) % NSET;
        issue(ra[nxt], rw[nxt], so);                                     // (past the end: zeros or values nobody uses; the
        read_tab(so);                                                    // table and the filter image are padded)
        bf16x8 a[NT][3], b[3];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            unsigned int p[3][4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                split3(ra[cur][i][c >> 1][2 * (c & 1)], ra[cur][i][c >> 1][2 * (c & 1) + 1], p[0][c], p[1][c], p[2][c]);
#pragma unroll
            for (int q = 0; q < 3; ++q) a[i][q] = __builtin_bit_cast(bf16x8, make_uint4(p[q][0], p[q][1], p[q][2], p[q][3]));
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) b[q] = __builtin_bit_cast(bf16x8, rw[cur][q]);
#pragma unroll
        for (int i = 0; i < NT; ++i) split_mfma2<NPROD>(a[i], b, acc[i], acc_lo[i]);
    };
    for (int s = sbeg; s < send; s += NSET) {
        step(s, std::integral_constant<int, 0>{});
        if (s + 1 < send) step(s + 1, std::integral_constant<int, 1>{});
        if (s + 2 < send) step(s + 2, std::integral_constant<int, 2>{});
        if (s + 3 < send) step(s + 3, std::integral_constant<int, 3>{});
    }

#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] += acc_lo[i][r];
    const float bv = (bias != nullptr && blockIdx.y == 0) ? bias[lr] : 0.f;
    const int ocol = lr * g.o_col_stride;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = wave * 32 * NT + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int ob = sh_o[m];
            if (ob >= 0) {
                float* c = out + (size_t)ob + ocol;
                if (use_atomic)
                    atomicAdd(c, acc[i][r] + bv);
                else
                    *c = acc[i][r] + bv;
            }
        }
}

// Tap groups: a filter "row" has 11 time taps.  Rows 2 r and 2 r + 1 give three groups of 8 slots -- taps 0-7 of each row and
// taps 8-10 of both (two slots multiply zeros) -- and a last odd row gives two.  slot -> (row, tap), row = -1 for a slot
// that multiplies a zero.  Forward: row = kh (21 rows, 32 groups), tap = kw.  Dgrad of parity par: row = kh' with
// kh = 2 kh' + par (11 / 10 rows, 17 / 15 groups), tap = 10 - kw (the gathered address grows as kw falls).
__host__ __device__ inline int tap_groups(int nrows) { return 3 * (nrows / 2) + 2 * (nrows & 1); }
__device__ __forceinline__ void tap_slot(int nrows, int gq, int j, int& row, int& tap) {
    const int pairs = nrows >> 1;
    if (gq < 3 * pairs) {
        const int pr = gq / 3, w = gq - pr * 3;
        if (w < 2) {
            row = 2 * pr + w;
            tap = j;
        } else {
            row = 2 * pr + (j >> 2);
            tap = 8 + (j & 3);
        }
    } else {
        row = nrows - 1;
        tap = gq == 3 * pairs ? j : (j < 4 ? 8 + j : 11);
    }
    if (tap > 10) row = -1;
}
// Tap table (the two 16-byte runs of a group start at slots 0 and 4) and the filter split into fragment order:
// Wp[step = (pair, group)][plane][n][k half = channel of the pair][8 slots] bf16, zero for unused slots; `spare` zero steps
// and table entries behind the last one (the pipeline reads ahead).
//   mode 0 (forward): n = oc, pair channel = ci, offset = row T1 + tap;          w[((n 32 + c) 21 + row) 11 + tap]
//   mode 1 + par (dgrad): n = ci, pair channel = oc, offset = (10 - row) TP + tap; w[((c 32 + n) 21 + 2 row + par) 11 + 10 - tap]
__global__ void conv2_split_prepare_kernel(const float* __restrict__ w, int mode, int nrows, int line, int spare,
                                           int* __restrict__ tap_off, unsigned int* __restrict__ Wp) {
    const int ng = tap_groups(nrows);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < (ng + spare) * 2) {
        const int gq = min(i >> 1, ng - 1);
        int row, tap;
        tap_slot(nrows, gq, (i & 1) * 4, row, tap);                       // (slots 0 and 4 of every group are real taps,
        if (row < 0) tap_slot(nrows, gq, 0, row, tap);                   // except the last group's second run)
        tap_off[i] = (mode == 0 ? row : 10 - row) * line + tap;
    }
    // one thread per (step, n, half, slot pair): two adjacent slots -> one dword per plane
    const int total = (16 * ng + spare) * 32 * 2 * 4;
    if (i >= total) return;
    const int jp = i & 3, half = (i >> 2) & 1, n = (i >> 3) & 31, step = i >> 8;
    const int cp = step / ng, gq = step - cp * ng;
    float v[2] = {0.f, 0.f};
    if (cp < 16) {
        const int c = 2 * cp + half;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int row, tap;
            tap_slot(nrows, gq, jp * 2 + e, row, tap);
            if (row >= 0)
                v[e] = mode == 0 ? w[((n * 32 + c) * 21 + row) * 11 + tap]
                                 : w[((c * 32 + n) * 21 + 2 * row + (mode - 1)) * 11 + 10 - tap];
        }
    }
    unsigned int p1, p2, p3;
    split3(v[0], v[1], p1, p2, p3);
    unsigned int* dst = Wp + ((size_t)step * 3 * 32 * 16 + (n * 16 + half * 8 + jp * 2)) / 2;
    dst[0] = p1;
    dst[(32 * 16) / 2] = p2;
    dst[2 * (32 * 16) / 2] = p3;
}

// d(out) (B, 32, 21, T) -> zero-bordered copy (B, 32, 41, T + 20): 10 rows and 10 columns of zeros on every side
// (one padded row per workgroup iteration: the row is decoded once with 32-bit arithmetic -- per element, three 64-bit
// divisions made this copy of 32 MB a 57 us kernel on the chain of the backward pass)
__global__ __launch_bounds__(256) void conv2_pad_dout_kernel(const float* __restrict__ dy, int B, int T,
                                                             float* __restrict__ dyp) {
    const int TP = T + 20, nrows = B * 32 * 41;
    for (int r = blockIdx.x; r < nrows; r += gridDim.x) {
        const int bc = r / 41, fo = r - bc * 41 - 10;
        float* dst = dyp + (size_t)r * TP;
        if (fo < 0 || fo >= 21) {
            for (int tc = threadIdx.x; tc < TP; tc += 256) dst[tc] = 0.f;
        } else {
            const float* src = dy + ((size_t)bc * 21 + fo) * T - 10;
            for (int tc = threadIdx.x; tc < TP; tc += 256) dst[tc] = (tc >= 10 && tc < T + 10) ? src[tc] : 0.f;
        }
    }
}

// DS2_CONV_SPLIT: 6 (default) / 9 partial products, 0 = the direct kernels of conv.hip (read per call: the tests switch it)
int conv_split_mode() {
    const char* e = getenv("DS2_CONV_SPLIT");
    const int v = e ? atoi(e) : 6;
    return (v == 6 || v == 9) ? v : 0;
}

// Work decomposition.  A wave owns 64 positions (two row tiles sharing the filter fragments) when that still gives every
// SIMD about two waves, else 32; when even those are fewer than the chip's 1024 SIMDs (a short minibatch), K is split over
// workgroups too in the data gradient (float atomics into a zeroed output).  Measured, B = 10, forward, ms (64 positions: 1 / 2 splits):
// T_in = 200 0.211 / 0.155, 300 0.217 / 0.187, 500 0.232 / 0.252, 830 0.343 / 0.391, 1501 0.559 / 0.633 -- at 830 the
// 1332 waves leave a SIMD with one or two of them and the kernel takes the time of two.
struct GatherPlan {
    int nt, tiles, per, nsplit;
};
// (the forward pass is never split over K: float atomics would make its output -- and with it every activation, eval included
// -- vary in the last bits from run to run)
GatherPlan plan_gather(int M, int nstep, bool may_split) {
    static const int force_ks = ds2_tune_env("DS2_CONV_SPLIT_KS") ? atoi(ds2_tune_env("DS2_CONV_SPLIT_KS")) : 0;
    static const int force_nt = ds2_tune_env("DS2_CONV_SPLIT_NT") ? atoi(ds2_tune_env("DS2_CONV_SPLIT_NT")) : 0;
    GatherPlan p;
    const int waves64 = ds2_cdiv(M, 64);
    p.nt = force_nt > 0 ? force_nt : (waves64 >= 1536 ? 2 : 1);   // (B = 10: T_in 500 0.217 / 0.252 ms with 32 / 64, 830 a tie, 1100 0.469 / 0.403)
    p.tiles = ds2_cdiv(M, 128 * p.nt);
    const int waves = p.tiles * 4;
    const int ks = !may_split ? 1 : (force_ks > 0 ? force_ks : (waves >= 1024 ? 1 : (waves >= 512 ? 2 : 3)));
    p.per = ds2_cdiv(nstep, ks);
    p.nsplit = ds2_cdiv(nstep, p.per);
    return p;
}

// `zeroed`: nullptr = never split K (forward); else *zeroed must already be true when this launch splits (the CALLER clears the
// output once, in front of its first launch, if ANY of its launches splits: the two row-parity launches of the data gradient
// pick their split separately -- 31 against 30 rows per batch element can put them on either side of a threshold -- and a
// fill issued by the second one would wipe the rows the first one has written).
// g1 != nullptr: a second problem on the same A and out in the same launch (both unsplit over K, same positions per wave)
void launch_gather(int mode, const float* A, unsigned int a_bytes, const int* tab, const unsigned int* Wp, unsigned int w_bytes,
                   const float* bias, float* out, unsigned long long o_elems, const GatherGeom& g, int nstep, bool* zeroed,
                   hipStream_t st, const GatherGeom* g1 = nullptr, const int* tab1 = nullptr, const unsigned int* Wp1 = nullptr,
                   unsigned int w_bytes1 = 0, int force_nt = 0) {
    GatherPlan pl = plan_gather(g.M, nstep, zeroed != nullptr);
    if (force_nt > 0) {
        pl.nt = force_nt;
        pl.tiles = ds2_cdiv(g.M, 128 * pl.nt);
    }
    const int nt = pl.nt, tiles0 = pl.tiles, per = pl.per, nsplit = pl.nsplit;
    const int tiles1 = g1 ? ds2_cdiv(g1->M, 128 * nt) : 0;
    const int tiles = g1 ? 8 * ((tiles0 >> 3) + (tiles1 >> 3) + ((tiles0 & 7) ? 1 : 0) + ((tiles1 & 7) ? 1 : 0)) : tiles0;
    if (nsplit > 1 && !*zeroed) {       // (not reached from ds2_conv2_dgrad_split, which fills up front; kept for other callers)
        (void)hipMemsetAsync(out, 0, o_elems * sizeof(float), st);
        *zeroed = true;
    }
    (void)o_elems;
    dim3 grid(tiles, nsplit), block(256);
    const int at = nsplit > 1 ? 1 : 0;
#define DS2_GATHER_GO(P_, N_)                                                                                            \
    hipLaunchKernelGGL((conv2_gather_kernel<P_, N_>), grid, block, 0, st, A, a_bytes, tab, Wp, w_bytes, bias, out, g, per, at, \
                       tiles0, tiles1, tab1, Wp1, w_bytes1, g1 ? *g1 : g)
    if (mode == 6 && nt == 2) DS2_GATHER_GO(6, 2);
    else if (mode == 6) DS2_GATHER_GO(6, 1);
    else if (nt == 2) DS2_GATHER_GO(9, 2);
    else DS2_GATHER_GO(9, 1);
#undef DS2_GATHER_GO
}

}  // namespace

// workspace of the split forms, in floats: the tap table(s) and the filter in fragment order (bf16, 3 planes)
size_t ds2_conv2_split_ws_floats() { return 1024 + (size_t)(16 * 32 + SPARE) * 3 * 32 * 16 / 2 + 4096; }

// conv2 forward through the gather-GEMM; returns 0 when it ran, 1 when the caller should use the direct kernels
// (switched off, or operands beyond the 2 GB a buffer descriptor covers).
int ds2_conv2_fwd_split(const float* in, const float* weight, const float* bias, int B, int t1, float* out, void* ws,
                        hipStream_t st) {
    const int mode = conv_split_mode();
    if (mode == 0) return 1;
    const int tout = t1 - 10;
    const unsigned long long a_bytes = 4ull * B * 32 * 61 * t1, o_elems = 1ull * B * 32 * 21 * tout;
    if (a_bytes >= 0x7FFFFFF0ull || o_elems >= 0x7FFFFFF0ull) return 1;
    constexpr int NG = 32, NP = 16;                                      // tap_groups(21)
    int* tap_off = reinterpret_cast<int*>(ws);
    unsigned int* Wp = reinterpret_cast<unsigned int*>(ws) + 1024;
    const int prep = (NP * NG + SPARE) * 32 * 2 * 4;
    hipLaunchKernelGGL(conv2_split_prepare_kernel, dim3(ds2_cdiv(prep, 256)), dim3(256), 0, st, weight, 0, 21, t1, SPARE,
                       tap_off, Wp);
    GatherGeom g;
    g.M = B * 21 * tout;
    g.ngroups = NG;
    g.npairs = NP;
    g.pair_stride = 2 * 61 * t1;
    g.D = 61 * t1;
    g.row_len = tout;
    g.rows_per_b = 21;
    g.a_base = 0;
    g.a_b_stride = 32 * 61 * t1;
    g.a_row_stride = 2 * t1;
    g.o_base = 0;
    g.o_b_stride = 32 * 21 * tout;
    g.o_row_stride = tout;
    g.o_col_stride = 21 * tout;
    g.frows = 0;
    const int nstep = NP * NG;
    const unsigned int w_bytes = (unsigned int)((NP * NG + SPARE) * 3 * 32 * 16 * 2);
    launch_gather(mode, in, (unsigned int)a_bytes, tap_off, Wp, w_bytes, bias, out, o_elems, g, nstep, nullptr, st);
    return 0;
}

// workspace of the split dgrad, in floats: two tap tables, two filter images (one per row parity), the zero-bordered d(out)
size_t ds2_conv2_dgrad_split_ws_floats(int B, int t1) {
    const size_t img0 = (size_t)(16 * tap_groups(11) + SPARE) * 3 * 32 * 16 / 2, img1 = (size_t)(16 * tap_groups(10) + SPARE) * 3 * 32 * 16 / 2;
    return 1024 + img0 + img1 + (size_t)B * 32 * 41 * (t1 + 10) + 64;
}

// conv2 data gradient through the gather-GEMM (two launches, one per input-row parity); 0 = ran, 1 = not selected
int ds2_conv2_dgrad_split(const float* d_out, const float* weight, int B, int t1, float* d_in, void* ws, hipStream_t st) {
    const int mode = conv_split_mode();
    if (mode == 0) return 1;
    // Stand-alone it beats the direct kernel from T_in ~ 600 up (B = 10, ms: 830 0.398 / 0.491, 1100 0.561 / 0.594, 1501 0.718 /
    // 0.774) and loses below (500 0.342 / 0.311, 200 0.267 / 0.194: three times the positions of the forward pass, half the
    // taps, a padded copy of d(out) to make).  Inside the training step it runs beside the side stream's bf16 GEMMs: always
    // on, 240-step runs of the B = 10 bin mix: 16.58 / 16.63 / 16.67 ms against 16.77 / 16.78 / 16.74 with the direct kernel.
    // DS2_CONV_SPLIT_DGRAD = 0 / 1 forces a choice (read per call: the tests switch it); default: from B * T1 >=
    // DS2_CONV_SPLIT_DGRAD_MIN input columns.  Round 4 (the weight-gradient kernels now leave the tail of the backward pass
    // earlier): same-box A/B of the whole B = 10 step, ms, threshold 3500 (the round-3 default: bins under 7 s on the direct
    // kernel) 15.24-15.32, never 15.26-15.28, always 15.15-15.18; thresholds 0 / 1000 / 2000 equal (15.02-15.08 on another
    // box): 1000.
    const char* on = getenv("DS2_CONV_SPLIT_DGRAD");
    static const int dgrad_min = ds2_tune_env("DS2_CONV_SPLIT_DGRAD_MIN") ? atoi(ds2_tune_env("DS2_CONV_SPLIT_DGRAD_MIN")) : 1000;
    if (on ? on[0] != '1' : (long)B * t1 < dgrad_min) return 1;
    const int T = t1 - 10, TP = T + 20;
    const unsigned long long a_bytes = 4ull * B * 32 * 41 * TP, o_elems = 1ull * B * 32 * 61 * t1;
    if (a_bytes >= 0x7FFFFFF0ull || o_elems >= 0x7FFFFFF0ull) return 1;
    const size_t img0 = (size_t)(16 * tap_groups(11) + SPARE) * 3 * 32 * 16 / 2, img1 = (size_t)(16 * tap_groups(10) + SPARE) * 3 * 32 * 16 / 2;
    int* tab = reinterpret_cast<int*>(ws);
    unsigned int* Wp[2] = {reinterpret_cast<unsigned int*>(ws) + 1024, reinterpret_cast<unsigned int*>(ws) + 1024 + img0};
    float* dyp = reinterpret_cast<float*>(ws) + 1024 + img0 + img1;
    hipLaunchKernelGGL(conv2_pad_dout_kernel, dim3(B * 32 * 41 < 8192 ? B * 32 * 41 : 8192), dim3(256), 0, st, d_out, B, T, dyp);
    // ONE decision about the zero fill for both row-parity launches, made before the first of them runs
    bool zeroed = false;
    for (int par = 0; par < 2; ++par)
        if (plan_gather(B * (par == 0 ? 31 : 30) * t1, 16 * tap_groups(par == 0 ? 11 : 10), true).nsplit > 1) zeroed = true;
    if (zeroed) (void)hipMemsetAsync(d_in, 0, o_elems * sizeof(float), st);
    GatherGeom gg[2];
    unsigned int wb[2];
    GatherPlan pp[2];
    for (int par = 0; par < 2; ++par) {
        const int nrows = par == 0 ? 11 : 10, ng = tap_groups(nrows), RH = par == 0 ? 31 : 30;
        const int prep = (16 * ng + SPARE) * 32 * 2 * 4;
        hipLaunchKernelGGL(conv2_split_prepare_kernel, dim3(ds2_cdiv(prep, 256)), dim3(256), 0, st, weight, 1 + par, nrows, TP,
                           SPARE, tab + 512 * par, Wp[par]);
        GatherGeom& g = gg[par];
        g.M = B * RH * t1;
        g.ngroups = ng;
        g.npairs = 16;
        g.pair_stride = 2 * 41 * TP;
        g.D = 41 * TP;
        g.row_len = t1;
        g.rows_per_b = RH;
        g.a_base = 0;                                                    // (the +10 rows / columns sit in the tap offsets)
        g.a_b_stride = 32 * 41 * TP;
        g.a_row_stride = TP;
        g.o_base = par * t1;
        g.o_b_stride = 32 * 61 * t1;
        g.o_row_stride = 2 * t1;
        g.o_col_stride = 61 * t1;
        // (DS2_CONV_DGRAD_ROWS = 0: every workgroup walks all tap groups, as until round 5 -- A/B timing and the test that holds
        // the two walks to the same bits; read per call)
        const char* rows_env = getenv("DS2_CONV_DGRAD_ROWS");
        const bool rows_off = rows_env && rows_env[0] == '0';
        pp[par] = plan_gather(g.M, 16 * ng, true);
        g.frows = (!rows_off && pp[par].nsplit == 1) ? nrows : 0;
        wb[par] = (unsigned int)((16 * ng + SPARE) * 3 * 32 * 16 * 2);
    }
    // both parities in ONE launch when neither splits K (DS2_CONV_DGRAD_MERGE = 0: two launches, as until round 5 -- A/B timing)
    const char* merge_env = getenv("DS2_CONV_DGRAD_MERGE");
    const bool merge_off = merge_env && merge_env[0] == '0';
    if (!merge_off && pp[0].nsplit == 1 && pp[1].nsplit == 1) {
        launch_gather(mode, dyp, (unsigned int)a_bytes, tab, Wp[0], wb[0], nullptr, d_in, o_elems, gg[0], 16 * gg[0].ngroups,
                      &zeroed, st, &gg[1], tab + 512, Wp[1], wb[1], pp[0].nt);
    } else {
        for (int par = 0; par < 2; ++par)
            launch_gather(mode, dyp, (unsigned int)a_bytes, tab + 512 * par, Wp[par], wb[par], nullptr, d_in, o_elems, gg[par],
                          16 * gg[par].ngroups, &zeroed, st);
    }
    return 0;
}

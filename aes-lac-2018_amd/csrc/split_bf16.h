// fp32 products on the bf16 matrix pipe: error-free three-way operand splitting (shared by gemm.hip and conv_split.hip).
//
// a = a1 + a2 + a3 with a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2), round-to-nearest-even at each step: bf16 has
// fp32's exponent range and 8 significant bits, so the three terms carry all 24 bits of a (|a2| <= 2^-8 |a|,
// |a3| <= 2^-16 |a|).  A product a b is the sum of nine partial products ai bj, each EXACT in the matrix instruction's fp32
// accumulator; the kernels add six of them or all nine.  The three left out, a2 b3 + a3 b2 + a3 b3, are bounded by
// 2^-23 |a b| in the worst case of both bounds; over 10^6 random products the six-product sum is off by at most 2^-24.2 |a b|
// and 5.8e-9 rms -- a correctly rounded fp32 multiply is off by up to 2^-24 and 2.5e-8 rms (tests/test_split_cpu.py).
#pragma once
#include "ds2_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// LDS image of an operand slab: [row][plane][16 k] bf16, row pitch 112 B = 3 planes x 32 B + 16: the 16-byte fragment reads of
// a 32-row MFMA operand (row = lane & 31, k half = lane >> 5) are conflict-free for ds_read_b128's four 16-lane groups.
constexpr int SPLIT_PITCH = 112;

__device__ __forceinline__ unsigned int pack_bf16(float lo, float hi) {
    unsigned int v;                             // round to nearest even, lo -> bits 15:0, hi -> bits 31:16
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v) : "v"(lo), "v"(hi));
    return v;
}
// two fp32 values -> three dwords of packed bf16 pairs (planes 1, 2, 3)
__device__ __forceinline__ void split3(float a, float b, unsigned int& p1, unsigned int& p2, unsigned int& p3) {
#if defined(DS2_GEMM_ABL) && (DS2_GEMM_ABL & 16)          // timing-only ablation (results WRONG): no split arithmetic
    p1 = __builtin_bit_cast(unsigned int, a);
    p2 = __builtin_bit_cast(unsigned int, b);
    p3 = p1 ^ p2;
    return;
#endif
    p1 = pack_bf16(a, b);
    float ra = a - __builtin_bit_cast(float, p1 << 16), rb = b - __builtin_bit_cast(float, p1 & 0xffff0000u);
    p2 = pack_bf16(ra, rb);
    ra -= __builtin_bit_cast(float, p2 << 16);
    rb -= __builtin_bit_cast(float, p2 & 0xffff0000u);
    p3 = pack_bf16(ra, rb);
}
// the six (or nine) partial products of one 32 x 32 x 16 step, into TWO accumulators (round 4): `hi` takes a1 b1, `lo` every
// product that carries a residual term.  One accumulator for all of them ABSORBS the small products once the running sum is
// large: with 4800 same-sign terms of order one the sum is ~10^4 (ulp 2^-10) while a1 b3 + a3 b1 contribute ~2^-16 per k --
// each 16-k instruction's worth is below half an ulp and is rounded away, every time (the adversarial test of
// tests/test_kernels_gpu.py: 1.8e-5 of the result lost, in the nine-product family too; the f32-input kernels lose 7.5e-6 on
// the same operands).  Among themselves the residual products are 2^-8 of the leading ones, so `lo` keeps 8 more bits of
// them; the two accumulators are added once, in the epilogue.  Same MFMA count, two independent chains.
template <int NPROD>
__device__ __forceinline__ void split_mfma2(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16& hi, f32x16& lo) {
#if defined(DS2_GEMM_ABL) && (DS2_GEMM_ABL & 8)           // timing-only ablation (results WRONG): one product of the six
    hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], hi, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[2], lo, 0, 0, 0);
    return;
#endif
    if (NPROD == 9) {
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[2], lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[2], lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[1], lo, 0, 0, 0);
    }
    lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], hi, 0, 0, 0);
}
// (one accumulator: the recurrence kernels' K = 800 per step)
template <int NPROD>
__device__ __forceinline__ f32x16 split_mfma(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 c) {
    if (NPROD == 9) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[1], c, 0, 0, 0);
    }
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
    return c;
}

"""The arithmetic behind the split-operand kernels (csrc/split_bf16.h), restated in numpy and checked on the CPU.

An fp32 value a is written a = a1 + a2 + a3 with a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2), each conversion
round-to-nearest-even (what v_cvt_pk_bf16_f32 does).  Claims the kernels rely on:
  (1) the split is error-free: a1 + a2 + a3 == a exactly, for every finite fp32 whose third term does not underflow;
  (2) |a2| <= 2^-8 |a| and |a3| <= 2^-16 |a| (so the left-out products a2 b3 + a3 b2 + a3 b3 are <= 2^-24 (1 + 2^-8) |a b| in
      the worst case of both bounds, and ~2^-29 |a b| typically);
  (3) every partial product ai bj has at most 16 significant bits: exact in an fp32 accumulator;
  (4) six partial products reproduce a b to a relative error below one fp32 rounding (2^-24)."""
import numpy as np


def bf16_rne(x):
    """float32 -> the nearest bf16 (ties to even), returned as float32."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3(a):
    a = np.asarray(a, dtype=np.float32)
    a1 = bf16_rne(a)
    r1 = (a - a1).astype(np.float32)                    # exact: a and a1 share the leading bits
    a2 = bf16_rne(r1)
    r2 = (r1 - a2).astype(np.float32)
    a3 = bf16_rne(r2)
    return a1, a2, a3


def _samples(n, seed):
    rng = np.random.default_rng(seed)
    mant = rng.standard_normal(n).astype(np.float32)
    expo = rng.integers(-60, 60, size=n)
    x = np.ldexp(mant, expo).astype(np.float32)
    edge = np.array([1.0, -1.0, 16777215.0, 1.0 + 2.0 ** -23, 0.1, 1e-20, 3e38, 2.0 ** -100, 255.0 / 256, 0.0], np.float32)
    return np.concatenate([x, edge])


def test_three_bf16_terms_carry_all_24_bits():
    a = _samples(200000, 1)
    a1, a2, a3 = split3(a)
    total = a1.astype(np.float64) + a2.astype(np.float64) + a3.astype(np.float64)
    assert np.array_equal(total, a.astype(np.float64))
    nz = a != 0
    assert np.all(np.abs(a2[nz]) <= 2.0 ** -8 * np.abs(a[nz]))
    assert np.all(np.abs(a3[nz]) <= 2.0 ** -16 * np.abs(a[nz]))


def test_partial_products_are_exact_in_fp32_and_six_of_them_are_enough():
    a, b = _samples(100000, 2), _samples(100000, 3)[::-1].copy()
    sa, sb = split3(a), split3(b)
    exact = a.astype(np.float64) * b.astype(np.float64)
    six = np.zeros_like(exact)
    for i, j in ((0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)):
        p64 = sa[i].astype(np.float64) * sb[j].astype(np.float64)
        p32 = (sa[i] * sb[j]).astype(np.float32)                      # fp32 multiply of two bf16 values
        ok = np.isfinite(p32) & (np.abs(p64) >= 2.0 ** -126)          # (normal range: no underflow of the product)
        assert np.array_equal(p32[ok].astype(np.float64), p64[ok])    # 8 x 8 significant bits: exact
        six += p64
    ok = (exact != 0) & np.isfinite(exact) & (np.abs(exact) > 1e-30) & (np.abs(exact) < 1e30)
    rel = np.abs(six[ok] - exact[ok]) / np.abs(exact[ok])
    assert float(rel.max()) <= 2.0 ** -24                             # below one fp32 rounding of the product
    assert float(np.median(rel)) <= 2.0 ** -27
    nine = sum(sa[i].astype(np.float64) * sb[j].astype(np.float64) for i in range(3) for j in range(3))
    assert np.array_equal(nine[ok], exact[ok])                        # all nine: the exact product

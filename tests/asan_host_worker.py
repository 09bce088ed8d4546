"""Child process of tests/test_host_asan_cpu.py: runs with libasan preloaded, loads tests/libds2host_asan.so (the library's
host-side C++ built with -fsanitize=address,undefined) through ctypes and fuzzes its entry points.

  ds2_edit_distance    random sequences (incl. empty) against a pure-Python Levenshtein
  ds2_ctc_beam_search  hypothesis: T <= 12, A <= 6, widths 1..32, linear and log input, zero probabilities, tight and exact
                       output capacities -- with a beam wide enough to hold every prefix the result must be the most
                       probable LABELLING by exhaustive enumeration of all A^T alignments (the reference has no beam
                       decoder, test.py:21: the enumeration is the specification); any width: the result's reported
                       log-probability never exceeds the enumerated optimum, labels are valid, offsets increase.

A sanitizer report aborts the process (non-zero exit); the parent asserts on the exit code and on 'ERROR: AddressSanitizer' /
'runtime error' in stderr.  Prints 'ASAN_WORKER_OK <n cases>' on success.
"""
import ctypes
import itertools
import math
import sys

import numpy as np
from hypothesis import HealthCheck, given, settings, strategies as st

LIB = ctypes.CDLL(sys.argv[1])
I32P = ctypes.POINTER(ctypes.c_int32)
LIB.ds2_edit_distance.restype = ctypes.c_int
LIB.ds2_edit_distance.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
LIB.ds2_ctc_beam_search.restype = ctypes.c_int
LIB.ds2_ctc_beam_search.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
LIB.ds2_last_error.restype = ctypes.c_char_p
LIB.ds2_version.restype = ctypes.c_int
NCASES = [0]


def py_edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


LIBC = ctypes.CDLL(None)
LIBC.malloc.restype = ctypes.c_void_p
LIBC.malloc.argtypes = [ctypes.c_size_t]
LIBC.free.argtypes = [ctypes.c_void_p]


class exact_copy(object):
    """A malloc'ed buffer of EXACTLY the array's size: numpy and ctypes arrays come from Python's pooled allocator, where an
    out-of-bounds access of the library lands in a neighbouring object and goes unnoticed; libc malloc is intercepted by the
    preloaded ASan runtime, so its redzones sit right behind the last element."""

    def __init__(self, arr):
        self.n = max(int(arr.nbytes), 1)
        self.ptr = LIBC.malloc(self.n)
        ctypes.memmove(self.ptr, arr.ctypes.data, arr.nbytes)

    def __del__(self):
        LIBC.free(self.ptr)


def edit_distance_fuzz(n=400):
    rng = np.random.default_rng(5)
    for _ in range(n):
        a = rng.integers(0, 4, size=rng.integers(0, 40)).astype(np.int32)
        b = rng.integers(0, 4, size=rng.integers(0, 40)).astype(np.int32)
        ba, bb = exact_copy(a), exact_copy(b)
        got = LIB.ds2_edit_distance(ba.ptr if len(a) else None, len(a), bb.ptr if len(b) else None, len(b))
        assert got == py_edit_distance(list(a), list(b)), (a, b, got)
        NCASES[0] += 1
    assert LIB.ds2_edit_distance(None, 3, None, 0) < 0 and b'bad argument' in LIB.ds2_last_error()


def best_labelling(probs, blank):
    t, a = probs.shape
    table = {}
    for path in itertools.product(range(a), repeat=t):
        p = 1.0
        for i, c in enumerate(path):
            p *= float(probs[i, c])
        if p == 0.0:
            continue
        lab, prev = [], None
        for c in path:
            if c != blank and c != prev:
                lab.append(c)
            prev = c
        table[tuple(lab)] = table.get(tuple(lab), 0.0) + p
    return table


def run_beam(probs, blank, width, log_input, cap):
    t, a = probs.shape
    src = np.ascontiguousarray(np.log(np.maximum(probs, 1e-300)) if log_input else probs, dtype=np.float32)
    buf = exact_copy(src)
    labels, offs = exact_copy(np.zeros(cap, np.int32)), exact_copy(np.zeros(cap, np.int32))   # exactly cap entries
    n = ctypes.c_int(-1)
    lp = ctypes.c_float(0.0)
    rc = LIB.ds2_ctc_beam_search(buf.ptr, t, a, blank, width, int(log_input), labels.ptr if cap else None,
                                 offs.ptr if cap else None, cap, ctypes.addressof(n), ctypes.addressof(lp))
    k = max(0, min(n.value, cap))
    as_list = lambda b: list(ctypes.cast(b.ptr, I32P)[:k])            # noqa: E731
    return rc, as_list(labels), as_list(offs), n.value, lp.value


@settings(max_examples=250, deadline=None, suppress_health_check=list(HealthCheck))
@given(st.integers(0, 12).flatmap(lambda t: st.tuples(st.just(t), st.integers(1, 6))).filter(lambda ta: ta[1] ** ta[0] <= 50000),
       st.integers(0, 2 ** 31 - 1), st.integers(1, 32), st.booleans(), st.integers(0, 3))
def beam_fuzz(ta, seed, width, log_input, sparsity):
    t, a = ta
    rng = np.random.default_rng(seed)
    blank = int(rng.integers(0, a))
    probs = rng.random((t, a)) ** (1 + 2 * sparsity)
    if sparsity and a > 1:
        probs[rng.random((t, a)) < 0.2 * sparsity] = 0.0             # exact zeros: -inf in the log domain
    probs[probs.sum(1) == 0, blank] = 1.0
    probs = (probs / np.maximum(probs.sum(1, keepdims=True), 1e-30)).astype(np.float32)
    table = best_labelling(probs.astype(np.float64), blank)
    best_p = max(table.values())
    rc, labels, offs, n, lp = run_beam(probs, blank, width, log_input, t)
    assert rc == 0, LIB.ds2_last_error()
    assert n == len(labels) <= t and all(0 <= c < a and c != blank for c in labels)
    assert all(0 <= o < max(t, 1) for o in offs) and all(y > x for x, y in zip(offs, offs[1:]))
    assert lp <= math.log(best_p) + 1e-3                            # never more than the optimum
    if tuple(labels) in table:                                       # reported log p = the labelling's own probability ...
        if width >= a ** t:                                          # ... exactly, when nothing was ever pruned
            assert abs(lp - math.log(table[tuple(labels)])) <= 2e-3 + 1e-4 * abs(lp), (lp, table[tuple(labels)])
    if width >= a ** t:                                              # the beam held every prefix: the exact optimum
        assert abs(math.log(table[tuple(labels)]) - math.log(best_p)) <= 1e-5, (labels, table)
    # an output buffer that is one label short is refused, and nothing is written past it
    if n > 0:
        rc2, _, _, n2, _ = run_beam(probs, blank, width, log_input, n - 1)
        assert rc2 < 0 and n2 == n
    NCASES[0] += 1


if __name__ == '__main__':
    assert LIB.ds2_version() > 0
    edit_distance_fuzz()
    beam_fuzz()
    print('ASAN_WORKER_OK %d' % NCASES[0])

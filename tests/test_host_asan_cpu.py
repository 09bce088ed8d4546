"""AddressSanitizer + UndefinedBehaviorSanitizer over the library's HOST-side C++ (csrc/api.hip, csrc/decode_host.hip:
error string, Levenshtein distance, CTC prefix beam search with its dynamic containers) -- CPU build only, never on the
GPU box's device.  The sanitized object is built by g++ (csrc/build.py::build_host_sanitized) and exercised in a child
python that has libasan preloaded (tests/asan_host_worker.py: random + hypothesis fuzz against exhaustive enumeration)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gcc_file(name):
    r = subprocess.run(['gcc', '-print-file-name=' + name], capture_output=True, text=True)
    path = r.stdout.strip()
    return path if r.returncode == 0 and os.path.isabs(path) and os.path.exists(path) else None


def test_host_cpp_under_asan_and_ubsan():
    asan = _gcc_file('libasan.so')
    if asan is None:
        pytest.skip('no libasan in this image')
    sys.path.insert(0, os.path.join(ROOT, 'aes-lac-2018_amd', 'csrc'))
    import build as csrc_build
    lib = csrc_build.build_host_sanitized()
    env = dict(os.environ, LD_PRELOAD=asan, PYTHONPATH=ROOT,
               ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:halt_on_error=1', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'asan_host_worker.py'), lib], capture_output=True,
                       text=True, env=env, timeout=600)
    assert 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert 'ASAN_WORKER_OK' in r.stdout and int(r.stdout.split('ASAN_WORKER_OK')[1].split()[0]) >= 500


def test_sanitizer_build_catches_an_overflow():
    """The harness itself: the same preload + a deliberate one-element over-read must be reported (guards against a silently
    inert sanitizer set-up -- e.g. the runtime not first in the link order)."""
    asan = _gcc_file('libasan.so')
    if asan is None:
        pytest.skip('no libasan in this image')
    sys.path.insert(0, os.path.join(ROOT, 'aes-lac-2018_amd', 'csrc'))
    import build as csrc_build
    lib = csrc_build.build_host_sanitized()
    # (buffers from libc malloc, which the preloaded runtime intercepts -- Python's own small-object pools have no redzones)
    code = ('import ctypes,sys; L=ctypes.CDLL(sys.argv[1]); C=ctypes.CDLL(None); C.malloc.restype=ctypes.c_void_p; '
            'C.malloc.argtypes=[ctypes.c_size_t]; L.ds2_edit_distance.argtypes=[ctypes.c_void_p,ctypes.c_int,'
            'ctypes.c_void_p,ctypes.c_int]; a=C.malloc(16); b=C.malloc(16); ctypes.memset(a,0,16); ctypes.memset(b,0,16); '
            'print(L.ds2_edit_distance(a,4,b,5))')                    # claims 5 elements of a 4-element array
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS='detect_leaks=0:halt_on_error=1')
    r = subprocess.run([sys.executable, '-c', code, lib], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and 'AddressSanitizer' in r.stderr, (r.returncode, r.stderr[-2000:])

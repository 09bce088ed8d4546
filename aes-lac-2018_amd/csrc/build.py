#!/usr/bin/env python
"""Build libds2hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python aes-lac-2018_amd/csrc/build.py [--force] [-v]

Each .hip file is compiled to an object (cached by mtime) and linked into
aes-lac-2018_amd/ds2hip/libds2hip.so.  The .so is git-ignored but travels to the GPU box.

A second library, libds2hip_faultinject.so, differs in ONE object: gru_persist.hip compiled with
-DDS2_FAULT_INJECT=1, which lets DS2_GRU_DBG=64 drop a workgroup's arrival (and the other ablation bits take effect).
tests/fault_inject_worker.py loads it (the bounded-spin / sticky-flag / fall-back test), and so does bench.py's
recurrence-floor leg (ablated launches of the shipped kernels' own source); the release library never reads that variable.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OUT = os.path.join(PKG, 'ds2hip', 'libds2hip.so')
OUT_FI = os.path.join(PKG, 'ds2hip', 'libds2hip_faultinject.so')
OBJ = os.path.join(HERE, 'build')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-I', os.path.join(ROOT, 'include'), '-I', HERE, '-I', OBJ,
         '-Wall', '-Wno-unused-function'] + os.environ.get('DS2_HIPCC_EXTRA', '').split()   # e.g. -DDS2_TIMING=1


def source_id():
    """sha256 over the library's sources (every .hip / .h of csrc/ and include/ds2hip.h, names and contents) -- the value
    ``ds2_build_id()`` of a library built from this tree returns.  ds2hip/lib.py computes the same digest when it loads the
    library and refuses a binary that was built from other sources (the .so files travel to the GPU box beside the sources
    and are cached by mtime here: nothing else ties a binary to the tree it is benchmarked with)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(HERE) if f.endswith('.hip') or f.endswith('.h'))
    for f in files + [os.path.join(ROOT, 'include', 'ds2hip.h')]:
        path = f if os.path.isabs(f) else os.path.join(HERE, f)
        h.update(os.path.basename(path).encode() + b'\0')
        with open(path, 'rb') as fh:
            h.update(fh.read())
        h.update(b'\0')
    return h.hexdigest()[:32]


def _stamp_build_id():
    """csrc/build/build_id.h = #define DS2_BUILD_ID "<source_id()>", rewritten only when the digest changes (api.hip, the one
    file that includes it, is then recompiled by the mtime rule)."""
    os.makedirs(OBJ, exist_ok=True)
    path = os.path.join(OBJ, 'build_id.h')
    text = '#define DS2_BUILD_ID "%s"\n' % source_id()
    if not os.path.exists(path) or open(path).read() != text:
        with open(path, 'w') as f:
            f.write(text)
    return path


def _newer(src, dst, deps):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in [src] + deps)


HOST_SOURCES = ('api.hip', 'decode_host.hip')          # host-only translation units (no device code, no HIP header)


def build_host_sanitized(out=None):
    """TEST INFRASTRUCTURE, CPU only: the host-side C++ of the library (error string / version, Levenshtein distance, CTC
    prefix beam search) compiled by g++ with AddressSanitizer + UndefinedBehaviorSanitizer into a stand-alone shared object
    (default tests/libds2host_asan.so).  tests/test_host_asan_cpu.py loads it in a child python with libasan preloaded and
    fuzzes it.  Never loaded by the product, never run on the GPU box's device (sanitizers are unavailable there)."""
    out = out or os.path.join(ROOT, 'tests', 'libds2host_asan.so')
    srcs = [os.path.join(HERE, f) for f in HOST_SOURCES]
    if not _newer(srcs[0], out, srcs[1:] + [os.path.join(HERE, 'ds2_host.h'), os.path.join(ROOT, 'include', 'ds2hip.h')]):
        return out
    cmd = [os.environ.get('CXX', 'g++'), '-x', 'c++', '-std=c++17', '-O1', '-g', '-fno-omit-frame-pointer',
           '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-shared', '-fPIC',
           '-I', os.path.join(ROOT, 'include'), '-I', HERE] + srcs + ['-o', out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('sanitizer build failed: %s\n%s\n%s' % (' '.join(cmd), r.stdout, r.stderr))
    return out


def build_gemm_variant(name, flags, src_name='gemm.hip'):
    """libds2hip_<name>.so with gemm.hip (or ``src_name``) recompiled with ``flags`` (tools/gemm_ablate.py)."""
    src = os.path.join(HERE, src_name)
    obj = os.path.join(OBJ, '%s_%s.o' % (src_name[:-4], name))
    out = os.path.join(PKG, 'ds2hip', 'libds2hip_%s.so' % name)
    r = subprocess.run([HIPCC] + FLAGS + ['-DDS2_ABLATION_BUILD=1'] + list(flags) + ['-c', src, '-o', obj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed:\n%s\n%s' % (r.stdout, r.stderr))
    objs = [os.path.join(OBJ, f[:-4] + '.o') for f in sorted(os.listdir(HERE)) if f.endswith('.hip')]
    objs = [obj if o.endswith(os.sep + src_name[:-4] + '.o') else o for o in objs]
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    return out


def build_variant(name, flags, force=False):
    """libds2hip_<name>.so = the release objects with gru_persist.hip recompiled with ``flags`` (tools: ``timing`` =
    -DDS2_TIMING=1 for tools/gru_sweep.py and tools/gru_phase_timing.py).  Call build() first."""
    src = os.path.join(HERE, 'gru_persist.hip')
    obj = os.path.join(OBJ, 'gru_persist_%s.o' % name)
    out = os.path.join(PKG, 'ds2hip', 'libds2hip_%s.so' % name)
    deps = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith('.h')] + [os.path.join(ROOT, 'include', 'ds2hip.h')]
    if force or _newer(src, obj, deps):
        r = subprocess.run([HIPCC] + FLAGS + list(flags) + ['-c', src, '-o', obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n%s\n%s' % (r.stdout, r.stderr))
    objs = [os.path.join(OBJ, f[:-4] + '.o') for f in sorted(os.listdir(HERE)) if f.endswith('.hip')]
    objs = [obj if o.endswith(os.sep + 'gru_persist.o') else o for o in objs]
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    return out


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    id_header = _stamp_build_id()
    srcs = sorted(f for f in os.listdir(HERE) if f.endswith('.hip'))
    deps = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith('.h')]
    deps.append(os.path.join(ROOT, 'include', 'ds2hip.h'))
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(HERE, s)
        obj = os.path.join(OBJ, s[:-4] + '.o')
        objs.append(obj)
        if force or _newer(src, obj, deps + ([id_header] if s == 'api.hip' else [])):
            jobs.append([HIPCC] + FLAGS + ['-c', src, '-o', obj])
    fi_src = os.path.join(HERE, 'gru_persist.hip')
    fi_obj = os.path.join(OBJ, 'gru_persist_faultinject.o')
    fi_new = force or _newer(fi_src, fi_obj, deps)
    if fi_new:
        jobs.append([HIPCC] + FLAGS + ['-DDS2_FAULT_INJECT=1', '-c', fi_src, '-o', fi_obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed: %s\n%s\n%s' % (' '.join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or not os.path.exists(OUT):
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs)
    if jobs or force or not os.path.exists(OUT_FI):
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT_FI] +
            [fi_obj if o.endswith(os.sep + 'gru_persist.o') else o for o in objs])
    # test infrastructure: the stand-in for an RCCL channel kernel (tests/test_coresidency_gpu.py)
    st_src, st_out = os.path.join(ROOT, 'tests', 'co_resident_kernel.hip'), os.path.join(ROOT, 'tests', 'libco_resident.so')
    # (only where the test tree exists and is writable: a packaged / installed tree must still build the product library)
    if os.path.exists(st_src) and os.access(os.path.dirname(st_src), os.W_OK) and (force or _newer(st_src, st_out, [])):
        run([HIPCC, '--offload-arch=gfx950', '-O2', '-shared', '-fPIC', '-o', st_out, st_src])
    return OUT


def build_tuning(force=False):
    """libds2hip_tuning.so: EVERY source compiled with -DDS2_TUNING=1, the build in which ds2_tune_env (ds2_common.h) reads the
    tuning knobs the release library ignores (tile widths, split-K targets, hand-off timing policies, forms no default dispatch
    reaches).  For tools/ sweeps: `DS2_LIB_VARIANT=tuning python tools/...`.  Objects under csrc/build/tuning/."""
    out_dir = os.path.join(OBJ, 'tuning')
    os.makedirs(out_dir, exist_ok=True)
    id_header = _stamp_build_id()
    srcs = sorted(f for f in os.listdir(HERE) if f.endswith('.hip'))
    deps = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith('.h')] + [os.path.join(ROOT, 'include', 'ds2hip.h')]
    jobs, objs = [], []
    for s in srcs:
        src, obj = os.path.join(HERE, s), os.path.join(out_dir, s[:-4] + '.o')
        objs.append(obj)
        if force or _newer(src, obj, deps + ([id_header] if s == 'api.hip' else [])):
            jobs.append([HIPCC] + FLAGS + ['-DDS2_TUNING=1', '-c', src, '-o', obj])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed: %s\n%s\n%s' % (' '.join(cmd), r.stdout, r.stderr))

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    out = os.path.join(PKG, 'ds2hip', 'libds2hip_tuning.so')
    if jobs or not os.path.exists(out):
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
    return out


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='-v' in sys.argv))
    if '--variant' in sys.argv and sys.argv[sys.argv.index('--variant') + 1] == 'tuning':
        print(build_tuning(force='--force' in sys.argv))
    elif '--variant' in sys.argv:
        name = sys.argv[sys.argv.index('--variant') + 1]
        flags = {'timing': ['-DDS2_TIMING=1']}.get(name)
        if flags is None:                      # e.g. --variant spec_12_4 -> -DDS2_SPEC_DELAY=12 -DDS2_SPEC_BACKOFF=4
            _, d, b = name.split('_')
            flags = ['-DDS2_SPEC_DELAY=%s' % d, '-DDS2_SPEC_BACKOFF=%s' % b]
        print(build_variant(name, flags))

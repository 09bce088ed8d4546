"""ctypes binding of libds2hip.so (the C ABI declared in include/ds2hip.h).

The library is the product: there is no CPU or PyTorch fallback.  Importing this module
without a built libds2hip.so raises; calling an entry point on tensors that are not on a
ROCm device raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libds2hip.so')
# DS2_LIB_VARIANT=<name>: load libds2hip_<name>.so instead -- the tools' way to a variant build (csrc/build.py --variant:
# `tuning` = the library that reads the tuning knobs, `timing`, `faultinject`, ablations); never set by the product or the tests
if os.environ.get('DS2_LIB_VARIANT'):
    LIB_PATH = os.path.join(_HERE, 'libds2hip_%s.so' % os.environ['DS2_LIB_VARIANT'])

_I = ctypes.c_int
_F = ctypes.c_float
_P = ctypes.c_void_p
_Z = ctypes.c_size_t

# name -> (restype, argtypes); tests/test_host_cpu.py checks this table against include/ds2hip.h and the .so
SIGNATURES = {
    'ds2_last_error': (ctypes.c_char_p, []),
    'ds2_version': (_I, []),
    'ds2_build_id': (ctypes.c_char_p, []),
    'ds2_spectrogram_ws_bytes': (_Z, [_I, _I]),
    'ds2_spectrogram_fwd': (_I, [_P, _P, _I, _I, _I, _F, _P, _P, _P]),
    'ds2_pcm16_to_float': (_I, [_P, _Z, _F, _P, _P]),
    'ds2_wsola_tempo': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    'ds2_gain_requantize': (_I, [_P, _P, _P, _I, _F, _P, _P]),
    'ds2_gemm_split_mode': (_I, [_I]),
    'ds2_gemm_f32': (_I, [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _F, _I, _P]),
    'ds2_gemm_f32_tn_group': (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'ds2_transpose_btf_to_bft': (_I, [_P, _I, _I, _I, _P, _P]),
    'ds2_conv_wt_ws_floats': (_Z, [_I]),
    'ds2_conv_fwd': (_I, [_I, _P, _P, _P, _I, _I, _P, _P, _P]),
    'ds2_conv2_dgrad_ws_floats': (_Z, [_I, _I]),
    'ds2_conv2_dgrad': (_I, [_P, _P, _I, _I, _P, _P, _Z, _P]),
    'ds2_conv_wgrad': (_I, [_I, _P, _P, _I, _I, _P, _P, _P]),
    'ds2_bn_ws_bytes': (_Z, [_I]),
    'ds2_bn2d_stats': (_I, [_P, _I, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P]),
    'ds2_bn2d_apply_htanh': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    'ds2_bn2d_htanh_bwd': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    'ds2_bn1d_stats': (_I, [_P, _P, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P]),
    'ds2_bn1d_apply': (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P]),
    'ds2_bn1d_bwd': (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    'ds2_gru_bidir_fwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    'ds2_gru_bidir_bwd': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'ds2_transpose2d': (_I, [_P, _I, _I, _P, _P]),
    'ds2_transpose2d_group': (_I, [_I, _P, _I, _I, _I, _P, _P]),
    'ds2_gru_sync_ws_bytes': (_Z, [_I, _I]),
    'ds2_gru_sync_error_offset': (_Z, []),
    'ds2_gru_persistent_supported': (_I, [_I, _I]),
    'ds2_gru_bidir_fwd_persistent': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'ds2_gru_bidir_bwd_persistent': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'ds2_gru_bidir_bwd_persistent_ex': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'ds2_gru_bwd_dh_supported': (_I, [_I, _I]),
    'ds2_gru_bidir_fwd_persistent_ex': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'ds2_gru_bwd_coef': (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    'ds2_gru_bidir_bwd_persistent_dh': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'ds2_softmax_rows': (_I, [_P, _I, _I, _P, _P]),
    'ds2_argmax_rows': (_I, [_P, _I, _I, _P, _P]),
    'ds2_greedy_collapse': (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P]),
    'ds2_ctc_ws_bytes': (_Z, [_I, _I, _I, _I]),
    'ds2_ctc_loss_grad': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P, _P, _P, _P]),
    'ds2_sumsq_ws_bytes': (_Z, [_Z]),
    'ds2_sumsq': (_I, [_P, _Z, _P, _P, _P]),
    'ds2_clip_sgd_nesterov': (_I, [_P, _P, _P, _Z, _P, _F, _F, _F, _F, _I, _P]),
    'ds2_step_stats': (_I, [_P, _I, _P, _P, _I, _P, _P]),
    'ds2_add2': (_I, [_P, _P, _Z, _P, _P]),
    'ds2_stream_create': (_I, [_I, _P]),
    'ds2_stream_destroy': (_I, [_P]),
    'ds2_edit_distance': (_I, [_P, _I, _P, _I]),
    'ds2_ctc_beam_search': (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P]),
}

ABI_VERSION = 402            # DS2_ABI_VERSION of include/ds2hip.h: the revision this table (and ops.py) is written against

_lib = None

ERR_ARG, ERR_LAUNCH, ERR_UNSUPPORTED = -1, -2, -3           # DS2_ERR_* of include/ds2hip.h


class Ds2Error(RuntimeError):
    """A non-zero status from an entry point; ``code`` is the DS2_ERR_* value."""

    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


def load():
    """Load libds2hip.so once; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libds2hip.so is missing at %s -- run `python aes-lac-2018_amd/csrc/build.py` '
                               '(or __graft_entry__.build()); there is no fallback path' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        # the revision FIRST, before any other symbol is resolved: a stale binary that lacks a newer entry point must die with
        # the 'rebuild it' message below, not with a bare AttributeError from the binding loop
        lib.ds2_version.restype = ctypes.c_int
        lib.ds2_version.argtypes = []
        have = lib.ds2_version()
        if have == ABI_VERSION:
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
        if have != ABI_VERSION:
            # signatures change between revisions without a change of symbol name: a stale binary would mis-pass arguments
            raise RuntimeError('%s is ABI revision %d, this binding is written against %d (include/ds2hip.h DS2_ABI_VERSION) '
                               '-- rebuild it: python aes-lac-2018_amd/csrc/build.py' % (LIB_PATH, have, ABI_VERSION))
        _check_build_id(lib)
        _lib = lib
    return _lib


def _check_build_id(lib):
    """The loaded binary must have been built from the sources beside it (csrc/build.py: source_id() is stamped into the
    library and recomputed here).  Skipped where there are no sources (a binary-only install) or with DS2_SKIP_BUILD_CHECK=1."""
    if os.environ.get('DS2_SKIP_BUILD_CHECK') == '1':
        return
    build_py = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'csrc', 'build.py')
    if not os.path.exists(build_py):
        return
    import importlib.util
    spec = importlib.util.spec_from_file_location('ds2_build_for_id', build_py)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    have, want = lib.ds2_build_id().decode(), mod.source_id()
    if have != want:
        raise RuntimeError('%s was built from other sources than the ones beside it (build id %s, tree %s) -- rebuild it: '
                           'python aes-lac-2018_amd/csrc/build.py' % (LIB_PATH, have, want))


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        if not x.is_cuda:
            raise RuntimeError('ds2hip entry points take device tensors; got a CPU tensor')
        if not x.is_contiguous():
            raise RuntimeError('ds2hip entry points take contiguous tensors')
        return x.data_ptr()
    return x


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_raw_device = getattr(torch._C, '_cuda_getDevice', None)


def stream_ptr():
    """The raw HIP stream torch considers current on the current device.  ``torch.cuda.current_stream().cuda_stream`` builds
    a Stream object and resolves the device index through three Python layers (~8 us; ~100 calls per training step, the
    first dozen of them with the GPU idle under the synchronise-per-step protocol); the two C entry points below are what it
    ends up calling."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


_fn_cache = {}


def call(name, *args):
    """Invoke an int-returning entry point on the current torch stream; raise on error."""
    fn = _fn_cache.get(name)
    if fn is None:
        fn = _fn_cache[name] = getattr(load(), name)
    rc = fn(*[a.data_ptr() if isinstance(a, torch.Tensor) and a.is_cuda and a.is_contiguous() else _ptr(a)
              for a in args], stream_ptr())
    if rc != 0:
        raise Ds2Error(rc, '%s failed (%d): %s' % (name, rc, load().ds2_last_error().decode()))


def host_call(name, *args):
    """Invoke a HOST entry point (numpy arrays / ctypes objects in, no stream); returns its int result, raising on a
    negative one."""
    import numpy as np
    conv = []
    for a in args:
        if isinstance(a, np.ndarray):
            if not a.flags['C_CONTIGUOUS']:
                raise RuntimeError('host entry points take C-contiguous arrays')
            conv.append(a.ctypes.data)
        elif isinstance(a, (ctypes.c_int, ctypes.c_float)):
            conv.append(ctypes.addressof(a))
        else:
            conv.append(a)
    rc = getattr(load(), name)(*conv)
    if rc < 0:
        raise RuntimeError('%s failed (%d): %s' % (name, rc, load().ds2_last_error().decode()))
    return rc


def query(name, *args):
    """Invoke a size query (no stream, no status)."""
    return getattr(load(), name)(*args)

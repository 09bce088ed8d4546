"""Per-time-step cost of the persistent BiGRU kernels, with the timing-only ablations of the DS2_TIMING build.

    python aes-lac-2018_amd/csrc/build.py --variant timing      # libds2hip_timing.so (DS2_GRU_DBG is honoured there only)
    python tools/gru_sweep.py [BSZ ...]                         # default 10

DS2_GRU_DBG bits (results are WRONG when set): 1 = do not wait for arrivals, 2 = skip the hand-off loads + MFMAs,
4 = skip the store drain.  Extra env (DS2_GRU_FWD_SPLIT, ...) passes through.
"""
# (the knobs this tool sweeps are TUNING knobs: read only by `python aes-lac-2018_amd/csrc/build.py --variant tuning`,
# i.e. run it with DS2_LIB_VARIANT=tuning -- the release library ignores them; csrc/ds2_common.h: ds2_tune_env)
import os
import sys

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(_ROOT, 'aes-lac-2018_amd'))
sys.path.insert(0, _ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ds2hip import lib  # noqa: E402

variant = os.environ.get('DS2_SWEEP_LIB', 'timing')
if variant:
    lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), 'libds2hip_%s.so' % variant)
from ds2hip import ops  # noqa: E402

T, HID = int(os.environ.get('T', '405')), 800


def measure(bsz, reps=6):
    torch.manual_seed(0)
    w_hh = ((torch.rand(2, 3 * HID, HID) * 2 - 1) / HID ** 0.5).cuda()
    w_hh_t = torch.stack([ops.transpose2d(w_hh[0], 3 * HID, HID), ops.transpose2d(w_hh[1], 3 * HID, HID)], 0)
    gates = 0.1 * torch.randn(T, bsz, 2, 3 * HID, device='cuda')
    d_out = 0.01 * torch.randn(T, bsz, HID, device='cuda')
    res = {'fwd': [], 'bwd': []}
    for _ in range(reps):
        g = gates.clone()
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        ghn, hout = ops.gru_bidir_fwd(g, w_hh, T, bsz, HID)
        e[1].record()
        ops.gru_bidir_bwd(g, ghn, hout, d_out, w_hh_t, T, bsz, HID)
        e[2].record()
        torch.cuda.synchronize()
        res['fwd'].append(e[0].elapsed_time(e[1]) * 1e3 / T)
        res['bwd'].append(e[1].elapsed_time(e[2]) * 1e3 / T)
    for ws in ops._sync_ws.values():       # ablations leave the counters dirty
        ws.zero_()
    return float(np.median(res['fwd'])), float(np.median(res['bwd']))


if __name__ == '__main__':
    sizes = [int(a) for a in sys.argv[1:]] or [10]
    dbgs = [int(v) for v in os.environ.get('DBGS', '0,4,1,5,2,3,7').split(',')]
    for bsz in sizes:
        for dbg in dbgs:
            os.environ['DS2_GRU_DBG'] = str(dbg)
            f, b = measure(bsz)
            print('B=%2d DBG=%d  fwd %.2f us/step  bwd %.2f us/step' % (bsz, dbg, f, b), flush=True)
    if variant == 'timing':
        import ctypes
        try:
            fn = lib.load().ds2_debug_read_retries
            fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
            buf = ctypes.c_uint(0)
            fn(ctypes.byref(buf), 1)
            print('speculative protocol: %d re-load rounds (per wave and step) in this process' % buf.value)
        except AttributeError:
            pass

"""A/B of persistent BiGRU kernel variants selected by environment variables: parity against the launch-per-step kernels
and stand-alone time per recurrence step (T = 405, H = 800).

    python tools/gru_variants.py [BSZ ...]          # default 10
    VARIANTS="name:ENV=V,ENV=V;name2:..." overrides the list below.
"""
import os
import sys

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(_ROOT, 'aes-lac-2018_amd'))
sys.path.insert(0, _ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ds2hip import lib  # noqa: E402

if os.environ.get('DS2_SWEEP_LIB'):
    lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), 'libds2hip_%s.so' % os.environ['DS2_SWEEP_LIB'])
from ds2hip import ops  # noqa: E402

T, HID = int(os.environ.get('T', '405')), 800
DEFAULT = ('base:;'
           'small2:DS2_GRU_SMALL_FWD=2,DS2_GRU_SMALL_BWD=2;'
           'small3:DS2_GRU_SMALL_FWD=3,DS2_GRU_SMALL_BWD=3')


def run(bsz, mode, reps):
    ops.GRU_MODE = mode
    torch.manual_seed(0)
    w_hh = ((torch.rand(2, 3 * HID, HID) * 2 - 1) / HID ** 0.5).cuda()
    w_hh_t = torch.stack([ops.transpose2d(w_hh[0], 3 * HID, HID), ops.transpose2d(w_hh[1], 3 * HID, HID)], 0)
    gates = torch.randn(T, bsz, 2, 3 * HID, device='cuda')
    d_out = 0.1 * torch.randn(T, bsz, HID, device='cuda')
    tf, tb, out = [], [], None
    for _ in range(reps):
        g = gates.clone()
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        ghn, hout = ops.gru_bidir_fwd(g, w_hh, T, bsz, HID)
        e[1].record()
        fwd = (g.clone(), ghn.clone(), hout.clone())
        e1b = torch.cuda.Event(enable_timing=True)
        e1b.record()
        ops.gru_bidir_bwd(g, ghn, hout, d_out, w_hh_t, T, bsz, HID)
        e[2].record()
        torch.cuda.synchronize()
        ops.check_async_errors()
        tf.append(e[0].elapsed_time(e[1]) * 1e3 / T)
        tb.append(e1b.elapsed_time(e[2]) * 1e3 / T)
        out = fwd + (g, ghn)
    return float(np.median(tf)), float(np.median(tb)), out


if __name__ == '__main__':
    sizes = [int(a) for a in sys.argv[1:]] or [10]
    variants = []
    for item in os.environ.get('VARIANTS', DEFAULT).split(';'):
        name, _, envs = item.partition(':')
        variants.append((name, dict(kv.split('=', 1) for kv in envs.split(',') if kv)))
    touched = sorted({k for _, e in variants for k in e})
    for bsz in sizes:
        _, _, ref = run(bsz, 'step', 1)
        for name, env in variants:
            for k in touched:
                os.environ.pop(k, None)
            os.environ.update(env)
            try:
                f, b, out = run(bsz, 'persistent', int(os.environ.get('REPS', '7')))
            except Exception as exc:       # noqa: BLE001 -- a variant that does not launch is a result, not a crash
                print('B=%2d %-10s FAILED: %s' % (bsz, name, str(exc)[:200]), flush=True)
                for ws in ops._sync_ws.values():
                    ws.zero_()
                ops._persistent_off.clear()
                continue
            errs = []
            for a, r in zip(out, ref):
                errs.append(float((a - r).abs().max()) / max(float(r.abs().max()), 1.0))
            ok = all(e <= tol for e, tol in zip(errs, (2e-5, 2e-5, 2e-5, 2e-4, 2e-4)))
            print('B=%2d %-10s fwd %.2f us/step  bwd %.2f us/step   parity %s (max rel %.1e)'
                  % (bsz, name, f, b, 'ok' if ok else 'FAIL', max(errs)), flush=True)

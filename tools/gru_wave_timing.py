"""Per-WAVE phase stamps of one workgroup of the 4x4x1 recurrence kernels (library built with -DDS2_TIMING=1:
   python aes-lac-2018_amd/csrc/build.py --variant timing).  WHICH=fwd|bwd picks the launch that is stamped last.

Stamps (s_memtime, lane 0 of every wave, steps 100..131 of workgroup (5, 0, 0)):
 0 step top   1 saved-activation loads issued   2 (sleep) hand-off loads issued   3 hand-off data validated
 4 MFMAs issued   5 partial sums written to LDS   6 after the pre-gate barrier   7 after the wait for the previous step's stores
 8 gate math done, hand-off stores issued   9 after the end-of-step barrier   10 saved-activation stores issued
"""
import ctypes, os, sys
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(_ROOT, 'aes-lac-2018_amd')); sys.path.insert(0, _ROOT)
import numpy as np, torch
from ds2hip import lib
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), 'libds2hip_timing.so')
from ds2hip import ops
t, bsz, hid = 405, int(os.environ.get('BSZ', '10')), 800
which = os.environ.get('WHICH', 'fwd')
torch.manual_seed(0)
w_hh = ((torch.rand(2, 3 * hid, hid) * 2 - 1) / hid ** 0.5).cuda()
w_hh_t = torch.stack([ops.transpose2d(w_hh[0], 3 * hid, hid), ops.transpose2d(w_hh[1], 3 * hid, hid)], 0)
gates = 0.1 * torch.randn(t, bsz, 2, 3 * hid, device='cuda'); d_out = 0.01 * torch.randn(t, bsz, hid, device='cuda')
rr = lib.load().ds2_debug_read_retries
rr.argtypes = [ctypes.c_void_p, ctypes.c_int]; rr.restype = ctypes.c_int
nretry = ctypes.c_uint(0)
times = []
for it in range(4):
    if it == 3:
        rr(ctypes.addressof(nretry), 1)            # reset: count the last pass only
    g = gates.clone(); torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(); ghn, hout, coef = ops.gru_bidir_fwd(g, w_hh, t, bsz, hid, want_coef=True); e[1].record()
    if which == 'bwd':
        ops.gru_bidir_bwd(g, ghn, hout, d_out, w_hh_t, t, bsz, hid, spare_cus=int(os.environ.get("SPARE_CUS", "-1")), coef=coef)
    e[2].record(); torch.cuda.synchronize()
    times.append(e[0].elapsed_time(e[1]) * 1e3 / t if which == 'fwd' else e[1].elapsed_time(e[2]) * 1e3 / t)
us_step = float(np.median(times))
rr(ctypes.addressof(nretry), 0)
print('re-load rounds in the last pass (all workgroups, both kernels if WHICH=bwd): %d = %.3f per wave and step (%d waves x %d steps)'
      % (nretry.value, nretry.value / (8.0 * 240 * t), 8 * 240, t))
buf = np.zeros(32 * 8 * 12, dtype=np.int64)
fn = lib.load().ds2_debug_read_wave_timing
fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
assert fn(buf.ctypes.data) == 0
tk = buf.reshape(32, 8, 12).astype(np.float64)[:, :, :11]
period = np.median(np.diff(tk[:, 0, 0]))                      # ticks per step (wave 0's step tops)
us = us_step / period
print('%s B=%d: %.2f us per step (events), %.0f ticks per step -> %.3f ns per tick' % (which, bsz, us_step, period, us * 1e3))
names = ['top', 'early loads issued', 'hand-off loads issued', 'hand-off validated', 'MFMAs issued', 'partials in LDS',
         'after barrier 1', 'old stores acked', 'stores issued', 'after barrier 2', 'saved stores issued']
ref = tk[:, :1, :1]                                            # wave 0's step top of the same step
rel = np.median(tk - ref, axis=0) * us                         # [wave][stamp] us since wave 0's step top
print('us since wave 0 step top (median over 31 steps); rows = stamps, columns = waves 0..7')
for i, n in enumerate(names):
    print('  %2d %-24s %s' % (i, n, ' '.join('%6.2f' % v for v in rel[:, i])))
print('phase durations per wave (us): stamp i -> i+1')
d = np.median(np.diff(tk, axis=2), axis=0) * us
for i in range(10):
    print('  %2d->%2d %-22s %s' % (i, i + 1, names[i + 1], ' '.join('%6.2f' % v for v in d[:, i])))

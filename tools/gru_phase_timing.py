"""Per-phase clock of one workgroup of the backward recurrence kernel (needs a library built with -DDS2_TIMING=1:
   DS2_HIPCC_EXTRA=-DDS2_TIMING=1 python aes-lac-2018_amd/csrc/build.py --force)."""
import ctypes, os, sys
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(_ROOT, 'aes-lac-2018_amd')); sys.path.insert(0, _ROOT)
import numpy as np, torch
from ds2hip import lib
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), 'libds2hip_timing.so')
from ds2hip import ops
t, bsz, hid = 405, int(os.environ.get('BSZ', '10')), 800
torch.manual_seed(0)
w_hh = ((torch.rand(2, 3 * hid, hid) * 2 - 1) / hid ** 0.5).cuda()
w_hh_t = torch.stack([ops.transpose2d(w_hh[0], 3 * hid, hid), ops.transpose2d(w_hh[1], 3 * hid, hid)], 0)
gates = 0.1 * torch.randn(t, bsz, 2, 3 * hid, device='cuda'); d_out = 0.01 * torch.randn(t, bsz, hid, device='cuda')
for _ in range(3):
    g = gates.clone(); ghn, hout = ops.gru_bidir_fwd(g, w_hh, t, bsz, hid)
    ops.gru_bidir_bwd(g, ghn, hout, d_out, w_hh_t, t, bsz, hid); torch.cuda.synchronize()
buf = np.zeros(32 * 8, dtype=np.int64)
fn = lib.load().ds2_debug_read_timing
fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
assert fn(buf.ctypes.data) == 0
tk = buf.reshape(32, 8).astype(np.float64)
names = ['step top (early loads issued) -> [counted protocol: poll done]', '[counted protocol: barrier after poll]',
         'first-attempt delay + hand-off loads + validation + MFMA + fold', 'barrier before gates',
         'LDS reduce + gate math + wait for old stores + hand-off stores issued', '[counted protocol: store drain]', 'barrier at the end of the step',
         '[arrival add] + saved stores -> next step top']
d = np.diff(np.concatenate([tk, np.roll(tk[:, :1], -1, axis=0)], axis=1), axis=1)[:-1]     # ticks per phase, 31 steps
tot = d.sum(1)
us_per_tick = float(os.environ.get('US_PER_STEP', '3.45')) / np.median(tot)     # s_memtime runs at the core clock (~2.36 GHz)
print('ticks per step (s_memtime): median %.1f; scaled to %.2f us per step' % (np.median(tot), np.median(tot) * us_per_tick))
for i, n in enumerate(names):
    print('  %-48s %6.1f ticks  %5.2f us' % (n, np.median(d[:, i]), np.median(d[:, i]) * us_per_tick))

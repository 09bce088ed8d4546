"""d(h) hand-off backward recurrence (gru_bwd_persistent6_kernel) against the launch-per-step kernels and the d(gh) hand-off
form, and its time per step: python tools/gru_dh_check.py [BSZ ...]   (TSTEPS, SPARE_CUS, DS2_GRU_BWD6_SPEC from the env)"""
# (the knobs this tool sweeps are TUNING knobs: read only by `python aes-lac-2018_amd/csrc/build.py --variant tuning`,
# i.e. run it with DS2_LIB_VARIANT=tuning -- the release library ignores them; csrc/ds2_common.h: ds2_tune_env)
import os, sys
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(_ROOT, 'aes-lac-2018_amd')); sys.path.insert(0, _ROOT)
import numpy as np, torch
from ds2hip import lib
if os.environ.get('DS2_LIB_VARIANT'):
    lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), 'libds2hip_%s.so' % os.environ['DS2_LIB_VARIANT'])
from ds2hip import ops

hid = int(os.environ.get('HID', '800'))
t = int(os.environ.get('TSTEPS', '405'))
check = os.environ.get('CHECK', '1') == '1'
for bsz in [int(a) for a in sys.argv[1:]] or [10]:
    torch.manual_seed(0)
    w_hh = ((torch.rand(2, 3 * hid, hid) * 2 - 1) / hid ** 0.5).cuda()
    w_hh_t = torch.stack([ops.transpose2d(w_hh[0], 3 * hid, hid), ops.transpose2d(w_hh[1], 3 * hid, hid)], 0)
    gi = torch.randn(t, bsz, 2, 3 * hid, device='cuda')
    d_out = 0.1 * torch.randn(t, bsz, hid, device='cuda')
    for spare in [int(x) for x in os.environ.get('SPARE_CUS', '0,52,82').split(',')]:
        out = {}
        forms = os.environ.get('FORMS', 'step,dgh,dh' if check else 'dgh,dh').split(',')
        for form in forms:
            ops.GRU_MODE = 'step' if form == 'step' else 'persistent'
            times = []
            for rep in range(1 if form == 'step' else 5):
                g = gi.clone()
                os.environ['DS2_GRU_BWD_DH'] = '1' if form == 'dh' else '0'
                ghn, hout, coef = ops.gru_bidir_fwd(g, w_hh, t, bsz, hid, want_coef=True)
                if form == 'dh' and os.environ.get('COEF_PASS') == '1':       # the elementwise pass instead of the forward kernel's own
                    coef = ops.gru_bwd_coef(g, ghn, hout, t, bsz, hid)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.gru_bidir_bwd(g, ghn, hout, d_out, w_hh_t, t, bsz, hid, spare_cus=spare, coef=coef)
                e1.record()
                torch.cuda.synchronize()
                ops.check_async_errors()
                times.append(e0.elapsed_time(e1) * 1e3 / t)
            out[form] = (g, ghn, float(np.median(times)))
        msg = 'B=%d T=%d spare=%d ' % (bsz, t, spare) + '  '.join('%s %.3f us/step' % (f, out[f][2]) for f in forms if f != 'step')
        if check and 'step' in forms:
            for name in [f for f in forms if f != 'step']:
                for a, b, what in ((out[name][0], out['step'][0], 'd(gi)'), (out[name][1], out['step'][1], 'd(ghn)')):
                    scale = max(float(b.abs().max()), 1.0)
                    msg += '  %s %s maxdiff %.2e' % (name, what, float((a - b).abs().max()) / scale)
        print(msg, flush=True)

import os, sys, subprocess, json
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(_ROOT, 'aes-lac-2018_amd')); sys.path.insert(0, _ROOT)
import torch, numpy as np
from ds2hip import lib
if os.environ.get('DS2_LIB_VARIANT'):          # libds2hip_<variant>.so (csrc/build.py --variant / build_variant)
    lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), 'libds2hip_%s.so' % os.environ['DS2_LIB_VARIANT'])
from ds2hip import ops
t, bsz, hid = int(os.environ.get('TSTEPS', '405')), int(os.environ.get('BSZ', '10')), 800
torch.manual_seed(0)
w_hh = ((torch.rand(2, 3*hid, hid)*2-1)/hid**0.5).cuda()
w_hh_t = torch.stack([ops.transpose2d(w_hh[0], 3*hid, hid), ops.transpose2d(w_hh[1], 3*hid, hid)], 0)
gates = 0.1*torch.randn(t, bsz, 2, 3*hid, device='cuda'); d_out = 0.01*torch.randn(t, bsz, hid, device='cuda')
spare = int(os.environ.get('SPARE_CUS', '-1'))      # CUs the backward launch leaves free (-1: the library's default, 52)
res = {'fwd': [], 'bwd': []}
for _ in range(6):
    g = gates.clone(); torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(); ghn, hout, coef = ops.gru_bidir_fwd(g, w_hh, t, bsz, hid, want_coef=True); e[1].record()
    ops.gru_bidir_bwd(g, ghn, hout, d_out, w_hh_t, t, bsz, hid, spare_cus=spare, coef=coef); e[2].record(); torch.cuda.synchronize()
    res['fwd'].append(e[0].elapsed_time(e[1])*1e3/t); res['bwd'].append(e[1].elapsed_time(e[2])*1e3/t)
print('T=%d ' % t, end='')
print('DBG=%s B=%d spare_cus=%d  fwd %.2f us/step  bwd %.2f us/step' % (os.environ.get('DS2_GRU_DBG','0'), bsz, spare, np.median(res['fwd']), np.median(res['bwd'])))

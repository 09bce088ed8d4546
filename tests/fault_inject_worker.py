#!/usr/bin/env python
"""Bounded spins, the sticky timeout flag and the fall-back to the per-step kernels, against the FAULT-INJECTION build
of the library (libds2hip_faultinject.so: gru_persist.hip compiled with -DDS2_FAULT_INJECT=1, where DS2_GRU_DBG=64
makes workgroup 0 lose its arrival of step 2).  Run as a fresh process by tests/test_kernels_gpu.py -- the release
library ignores DS2_GRU_DBG, so this cannot be tested through it.

    python tests/fault_inject_worker.py fwd|bwd
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'aes-lac-2018_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    which = sys.argv[1]
    from ds2hip import lib
    lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), 'libds2hip_faultinject.so')
    from ds2hip import ops
    dev = torch.device('cuda')
    t, bsz, hid = 6, 10, 800
    torch.manual_seed(0)
    w_hh = (torch.randn(2, 3 * hid, hid) * 0.02).to(dev)
    w_hh_t = torch.stack([ops.transpose2d(w_hh[0], 3 * hid, hid), ops.transpose2d(w_hh[1], 3 * hid, hid)], 0)
    gi = torch.randn(t, bsz, 2, 3 * hid).to(dev)
    d_out = torch.randn(t, bsz, hid).to(dev)
    ops.GRU_MODE = 'auto'
    assert ops._use_persistent(dev, bsz, hid)

    def fwd():
        g = gi.clone()
        ghn, hout = ops.gru_bidir_fwd(g, w_hh, t, bsz, hid)
        return g, ghn, hout

    def bwd(saved):
        g, ghn, hout = (x.clone() for x in saved)
        ops.gru_bidir_bwd(g, ghn, hout, d_out, w_hh_t, t, bsz, hid)
        return g, ghn

    os.environ['DS2_GRU_DBG'] = '0'
    good_f = fwd()
    good_b = bwd(good_f)
    torch.cuda.synchronize()
    ops.check_async_errors()

    os.environ['DS2_GRU_DBG'] = '64'                       # workgroup 0 loses an arrival: every waiter must time out
    t0 = time.time()
    fwd() if which == 'fwd' else bwd(good_f)
    torch.cuda.synchronize()
    took = time.time() - t0
    assert 1.0 < took < 60.0, took             # (the fault-injection build's spins time out after 1.5 s; the release library's after 5)
    os.environ['DS2_GRU_DBG'] = '0'
    # a LATER launch must not erase the flag (round-1 bug: the per-launch memset cleared it): a training step makes ten
    # launches and checks once
    fwd()
    torch.cuda.synchronize()
    try:
        ops.check_async_errors()
    except RuntimeError as e:
        assert 'timed out' in str(e), e
    else:
        raise AssertionError('the timeout flag was lost')
    # the device now runs the per-step kernels, in the same process, and is correct
    assert not ops._use_persistent(dev, bsz, hid)
    f2 = fwd()
    b2 = bwd(good_f)
    torch.cuda.synchronize()
    ops.check_async_errors()
    for a, b in zip(f2 + b2, good_f + good_b):
        assert float((a - b).abs().max()) < 2e-5, float((a - b).abs().max())
    # and after the host-side reset the persistent kernels work again (the workspace was re-zeroed)
    ops._persistent_off.clear()
    f3 = fwd()
    b3 = bwd(good_f)
    torch.cuda.synchronize()
    ops.check_async_errors()
    for a, b in zip(f3 + b3, good_f + good_b):
        assert float((a - b).abs().max()) < 1e-6
    print('OK', which, '%.1f s' % took)


if __name__ == '__main__':
    main()

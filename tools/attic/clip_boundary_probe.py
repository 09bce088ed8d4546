"""Run-to-run reproducibility of a full trainer step, tensor by tensor through the conv block's backward pass.

NRUN fresh models (same seed) take a warm-up step and two steps on tests/test_coresidency_gpu.py's batches; every input and
output of the conv block's backward kernels in the first measured step is kept and compared with run 0's.  What it found
(round 3): two discrete outcomes for the second loss (267.1720 / 267.1837, B = 10, seed 10), in both recurrence kernel
families, caused by ONE element of conv2's output whose BatchNorm output is -1.4e-8 .. -7.9e-8: the last bits of the
batch statistics (atomic sums) and of the weights after the warm-up step decide on which side of the clip boundary it
falls, and with it whether a gradient of 0.067 (the largest is 0.14) passes.  Not a race: inputs of the BN backward kernel
are identical to 1e-4 in every run, the flipped element is always the same one, and its normalised value is printed below.
"""
import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/aes-lac-2018_amd')
import numpy as np, torch
from tests.test_coresidency_gpu import _batches
from codes.engine import Trainer
from codes.model import DeepSpeech
from ds2hip import ops
LOG = None
def wrap(name):
    orig = getattr(ops, name)
    def f(*a, **k):
        out = orig(*a, **k)
        if LOG is not None:
            for i, v in enumerate(a):
                if torch.is_tensor(v) and v.is_cuda:
                    LOG.append(('%s.in%d' % (name, i), v.detach().clone()))
            if torch.is_tensor(out):
                LOG.append(('%s.out' % name, out.detach().clone()))
        return out
    setattr(ops, name, f)
for n in ('transpose2d', 'bn2d_htanh_bwd', 'conv2_dgrad', 'conv_wgrad'):
    wrap(n)
def run(mode):
    global LOG
    ops.GRU_MODE = mode
    torch.manual_seed(7)
    model = DeepSpeech().to('cuda')
    opt = torch.optim.SGD(model.parameters(), lr=3e-4, momentum=0.9, nesterov=True)
    tr = Trainer(model, opt, device='cuda', max_norm=400)
    b = _batches(10, 10)
    tr.update(b[0]); torch.cuda.synchronize()
    LOG = []
    l0 = tr.update(b[0]); torch.cuda.synchronize()
    log, LOG = LOG, None
    l1 = tr.update(b[1]); torch.cuda.synchronize()
    return [l0, l1], log
res = [run('persistent') for _ in range(int(os.environ.get('NRUN', '12')))]
for i, (l, log) in enumerate(res):
    msg = ''
    for (n, v), (_, r) in zip(log, res[0][1]):
        e = float((v - r).abs().max()); s = float(r.abs().max()) + 1e-30
        if e > 1e-4 * s:
            nbad = int(((v - r).abs() > 1e-4 * s).sum())
            msg += ' %s %.2g/%.2g(n=%d)' % (n, e, s, nbad)
    print(i, ['%.6f' % v for v in l], msg)
# the flipped element: where does conv2's BN output sit relative to the clip boundaries?
good = dict((n, v) for n, v in res[0][1][:12])
for i, (l, log) in enumerate(res):
    d = dict()
    for n, v in log:
        d.setdefault(n, v)
    e = (d['bn2d_htanh_bwd.out'] - res[0][1][[n for n, _ in res[0][1]].index('bn2d_htanh_bwd.out')][1]).abs()
    if float(e.max()) > 1e-3:
        idx = int(e.argmax()); y2 = d['bn2d_htanh_bwd.in0']; c = (idx // (y2.shape[2] * y2.shape[3])) % y2.shape[1]
        mi = d['bn2d_htanh_bwd.in2'].double(); w = d['bn2d_htanh_bwd.in3'].double(); b = d['bn2d_htanh_bwd.in4'].double()
        x = y2.reshape(-1)[idx].double()
        for (m, s) in ((mi[c], mi[32 + c]), (mi[2 * c], mi[2 * c + 1])):
            print(i, 'element', idx, 'channel', c, 'bn output %.3e' % float((x - m) * s * w[c] + b[c]))

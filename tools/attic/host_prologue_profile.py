#!/usr/bin/env python
"""Host-side timeline of a training step with a synchronisation at its end (bench.py's headline protocol): when, after the
previous step's readback returned, does the host issue each library call?  Until the first long kernel is enqueued the GPU
idles behind the host, so every microsecond in front of it is step time.

    python tools/host_prologue_profile.py [steps=30] [ncalls=40]
"""
import os, sys, time
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(_ROOT, 'aes-lac-2018_amd')); sys.path.insert(0, _ROOT)
import numpy as np, torch
import bench
from ds2hip import lib
from codes.engine import Trainer
from codes.model import DeepSpeech
from codes.transforms import BatchSpectrogram

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ncalls = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device('cuda', 0)
plan = bench.bin_plan(10, 8)
mine = [bench.make_bin(p) for p in plan]
resident = [bench.make_resident(b, dev) for b in mine]
torch.manual_seed(42)
model = DeepSpeech().to(dev)
opt = torch.optim.SGD(model.parameters(), lr=3e-4, momentum=0.9, nesterov=True)
trainer = Trainer(model, opt, device=dev, max_norm=400)
front = BatchSpectrogram(device=dev)

log = []
orig_call = lib.call


def traced(name, *args):
    log.append((name, time.perf_counter()))
    return orig_call(name, *args)


def step(i):
    flat, offs, labels, lens = resident[i % len(resident)]
    inputs, pct = front(flat, offs)
    log.append(('<frontend returned>', time.perf_counter()))
    return trainer.update((inputs, labels, pct, lens), defer=False)


for i in range(8):
    step(i)
torch.cuda.synchronize()
lib.call = traced
import ds2hip.ops as ops_mod
for mod in list(sys.modules.values()):                       # modules that did ``from ds2hip.lib import call``
    if getattr(mod, 'call', None) is orig_call:
        mod.call = traced
rows, ends, walls = [], [], []
for i in range(8, 8 + steps):
    del log[:]
    t0 = time.perf_counter()
    step(i)
    t1 = time.perf_counter()
    rows.append([(n, (t - t0) * 1e6) for n, t in log])
    walls.append((t1 - t0) * 1e3)
print('step wall %.2f ms (median of %d); calls per step %d' % (float(np.median(walls)), steps, len(rows[0])))
print('host time since the step began (us, median over steps) at each of the first %d calls:' % ncalls)
for k in range(min(ncalls, min(len(r) for r in rows))):
    ts = [r[k][1] for r in rows]
    print('  %3d  %8.1f  %s' % (k, float(np.median(ts)), rows[0][k][0]))
last = [r[-1][1] for r in rows]
print('last call of the step issued at %.1f us (median): %s' % (float(np.median(last)), rows[0][-1][0]))

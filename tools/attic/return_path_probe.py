#!/usr/bin/env python
"""How long after the GPU has finished a training step does the host hold the step's loss?  (The window the GPU then idles
through, plus the host's first launches of the next step, is what a synchronisation per step costs.)

An event recorded right behind the step's last launch gives the GPU-side completion time on the reference event's clock; the
host's clock is tied to it by one synchronisation at the start."""
import os, sys, time
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(_ROOT, 'aes-lac-2018_amd')); sys.path.insert(0, _ROOT)
import numpy as np, torch
import bench
from codes.engine import Trainer
from codes.model import DeepSpeech
from codes.transforms import BatchSpectrogram

dev = torch.device('cuda', 0)
plan = bench.bin_plan(10, 8)
mine = [bench.make_bin(p) for p in plan]
resident = [bench.make_resident(b, dev) for b in mine]
torch.manual_seed(42)
model = DeepSpeech().to(dev)
trainer = Trainer(model, torch.optim.SGD(model.parameters(), lr=3e-4, momentum=0.9, nesterov=True), device=dev, max_norm=400)
front = BatchSpectrogram(device=dev)


def launch(i):
    flat, offs, labels, lens = resident[i % len(resident)]
    inputs, pct = front(flat, offs)
    return trainer.update((inputs, labels, pct, lens), defer=True)


for i in range(8):
    launch(i).result()
torch.cuda.synchronize()
ref = torch.cuda.Event(enable_timing=True)
ref.record()
ref.synchronize()
h_ref = time.perf_counter()
lat, first = [], []
for i in range(8, 48):
    t_start = time.perf_counter()
    pend = launch(i)
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    t_enq = time.perf_counter()
    pend.result()
    t_have = time.perf_counter()
    ev.synchronize()
    g_done = ref.elapsed_time(ev) * 1e-3                 # seconds since ref on the GPU's clock
    lat.append((t_have - h_ref - g_done) * 1e6)
    first.append((t_enq - t_start) * 1e6)
print('host holds the loss %.1f us (median; min %.1f, max %.1f) after the GPU finished the step' % (
    float(np.median(lat)), min(lat), max(lat)))
print('host time to enqueue a whole step: %.0f us (median)' % float(np.median(first)))

"""How much a training step slows down beside the RCCL stand-in (tests/co_resident_kernel.hip: 32 long-lived workgroups streaming
HBM on a third stream) -- the number DESIGN.md section 5's scaling prediction uses for 'interference'."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'aes-lac-2018_amd'))
import numpy as np, torch
import bench
from codes.engine import Trainer
from codes.model import DeepSpeech
from codes.transforms import BatchSpectrogram
from tests.test_kernels_gpu import co_resident_load
dev = torch.device('cuda', 0)
bsz = int(os.environ.get('BSZ', '10'))
plan = bench.bin_plan(bsz, bench.NUM_BINS, world=1)
res = [bench.make_resident(bench.make_bin(p), dev) for p in plan]
torch.manual_seed(42)
model = DeepSpeech().to(dev)
opt = torch.optim.SGD(model.parameters(), lr=3e-4, momentum=0.9, nesterov=True)
trainer = Trainer(model, opt, device=dev, max_norm=400)
front = BatchSpectrogram(device=dev)
def step(i):
    flat, offs, labels, lens = res[i % len(res)]
    inputs, pct = front(flat, offs)
    return trainer.update((inputs, labels, pct, lens), defer=False)
for i in range(24): step(i)
def timed(n=24):
    torch.cuda.current_stream().synchronize(); t0 = time.time()
    for i in range(n): step(i)
    torch.cuda.current_stream().synchronize(); return (time.time() - t0) / n * 1e3
alone = timed()
held = co_resident_load(duration_ms=2000.0)
beside = timed()
still = not held[1].query()
held[1].synchronize()
print('B=%d: %.2f ms per step alone, %.2f ms beside the 32-workgroup HBM-streaming stand-in (%+.1f %%; stand-in still running at the end: %s)'
      % (bsz, alone, beside, 100.0 * (beside / alone - 1.0), still))

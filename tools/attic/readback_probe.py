"""How long does the host take to see the end of a step?  One tiny launch (ds2_step_stats) + the readback, repeated:
(a) device buffer -> non-blocking copy into page-locked memory -> event -> polled with event.query() (the trainer's path);
(b) the kernel writes straight into page-locked host memory and the host polls the memory."""
import os, sys, time, ctypes
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(_ROOT, 'aes-lac-2018_amd')); sys.path.insert(0, _ROOT)
import numpy as np, torch
from ds2hip import lib, ops
dev = torch.device('cuda', 0)
costs = torch.ones(10, device=dev)
sumsq = torch.ones(1, dtype=torch.float64, device=dev)
stats = torch.empty(4, dtype=torch.float64, device=dev)
slot = torch.empty(4, dtype=torch.float64).pin_memory()
N = 300
def ev_mode():
    costs.add_(1.0)
    ops.step_stats(costs, sumsq, stats)
    slot.copy_(stats, non_blocking=True)
    e = torch.cuda.Event(); e.record()
    ops.spin_wait(e)
    return slot[0].item()
host = torch.zeros(8, dtype=torch.float64).pin_memory()
hnp = host.numpy()
words = ops.async_error_words()
table = torch.tensor([w.data_ptr() for w in words] or [0], dtype=torch.int64).to(dev)
def direct_mode():
    costs.add_(1.0)
    hnp[0] = -1.0
    lib.call('ds2_step_stats', costs, costs.numel(), sumsq, table, len(words), host.data_ptr())
    while hnp[0] < 0: pass
    return hnp[0]
for name, fn in (('event', ev_mode), ('direct', direct_mode), ('event', ev_mode), ('direct', direct_mode)):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): v = fn()
    dt = (time.perf_counter() - t0) / N
    print('%-7s %.1f us per launch+readback (last value %.0f)' % (name, dt * 1e6, v))

import os, sys
sys.path.insert(0, 'aes-lac-2018_amd'); sys.path.insert(0, '.')
import torch, numpy as np
from ds2hip import ops
for T, B, A, L in [(391, 10, 29, 120), (746, 10, 29, 250), (496, 32, 29, 160), (746, 8, 43, 250)]:
    rng = np.random.default_rng(0)
    acts = torch.from_numpy((2 * rng.standard_normal((T, B, A))).astype(np.float32)).cuda()
    lens = [L - (i % 5) * 7 for i in range(B)]
    labels = torch.from_numpy(rng.integers(1, A, size=sum(lens)).astype(np.int32)).cuda()
    offs = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)).cuda()
    ll = torch.tensor(lens, dtype=torch.int32).cuda(); al = torch.tensor([T - (i % 3) * 11 for i in range(B)], dtype=torch.int32).cuda()
    for _ in range(3): c, g = ops.ctc_loss_grad(acts, labels, offs, ll, al, max(lens))
    torch.cuda.synchronize(); ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); c, g = ops.ctc_loss_grad(acts, labels, offs, ll, al, max(lens)); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print('T=%d B=%d A=%d L=%d: %.1f us (3 kernels)  cost0 %.4f' % (T, B, A, L, np.median(ts), float(c[0])))

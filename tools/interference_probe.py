"""What slows the persistent recurrence kernels when other work runs beside them on the CUs they leave free?
Times a stand-alone BiGRU layer pass (T = 405, B = 10, H = 800) on the main stream while a low-priority side stream runs
(a) nothing, (b) a cache-resident fp32 matmul (matrix pipes + power, little memory traffic), (c) a streaming copy,
(d) a streaming read (sum), (e) the layer's real weight-gradient GEMM (TN, split-K with atomics), (f) the same without
split-K (no atomics)."""
import os, sys
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(_ROOT, 'aes-lac-2018_amd')); sys.path.insert(0, _ROOT)
import torch, numpy as np
from ds2hip import ops
t, bsz, hid = 405, int(os.environ.get('BSZ', '10')), 800
rows = t * bsz
torch.manual_seed(0)
dev = torch.device('cuda')
w_hh = ((torch.rand(2, 3 * hid, hid) * 2 - 1) / hid ** 0.5).cuda()
w_hh_t = torch.stack([ops.transpose2d(w_hh[0], 3 * hid, hid), ops.transpose2d(w_hh[1], 3 * hid, hid)], 0)
gates = 0.1 * torch.randn(t, bsz, 2, 3 * hid, device=dev); d_out = 0.01 * torch.randn(t, bsz, hid, device=dev)
side = ops.low_priority_stream(dev)
ma, mb = torch.randn(2048, 2048, device=dev), torch.randn(2048, 2048, device=dev)
big = torch.randn(64 << 20, device=dev); big2 = torch.empty_like(big)
dgi = torch.randn(rows, 6 * hid, device=dev); xin = torch.randn(rows, hid, device=dev); gw = torch.empty(6 * hid, hid, device=dev)


def beside(kind):
    if kind == 'matmul':
        for _ in range(12): torch.mm(ma, mb)
    elif kind == 'copy':
        for _ in range(8): big2.copy_(big)
    elif kind == 'sum':
        for _ in range(16): big.sum()
    elif kind == 'dw':
        for _ in range(3): ops.gemm(dgi, xin, trans_a=True, out=gw, split_k=0)
    elif kind == 'dw_nosplit':
        for _ in range(2): ops.gemm(dgi, xin, trans_a=True, out=gw, split_k=1)


big_a, big_b = torch.randn(8192, 512, device=dev), torch.randn(512, 8192, device=dev)      # 8192 x 8192 x 512: 268 MB of C per call


def beside2(kind):
    if kind == 'mm_stream':
        for _ in range(6): torch.mm(big_a, big_b)
    else:
        beside(kind); beside(kind)


# the recurrence is launched FIRST (as in the training step: its workgroups are resident before the side work arrives)
for kind in ('none', 'matmul', 'mm_stream', 'copy', 'sum', 'dw', 'dw_nosplit', 'none'):
    res = {'both': [], 'side': []}
    for _ in range(5):
        g = gates.clone(); torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record(); ghn, hout = ops.gru_bidir_fwd(g, w_hh, t, bsz, hid)
        with torch.cuda.stream(side):
            e[2].record(side); beside2(kind); e[3].record(side)
        ops.gru_bidir_bwd(g, ghn, hout, d_out, w_hh_t, t, bsz, hid); e[1].record(); torch.cuda.synchronize()
        res['both'].append(e[0].elapsed_time(e[1]) * 1e3 / (2 * t)); res['side'].append(e[2].elapsed_time(e[3]))
    print('%-11s fwd+bwd %.2f us/step  (side work %.2f ms beside %.2f ms of recurrence)'
          % (kind, np.median(res['both']), np.median(res['side']), np.median(res['both']) * 2 * t / 1e3), flush=True)

"""Stand-alone timing of the BatchNorm kernels on the recurrent stack's shapes (rows = T*B = 4050, 800 features, input = the sum
of the two directions' outputs) and the conv
block's (B = 10, 32 channels, 61 x 415 and 21 x 405): microseconds and the fraction of 8 TB/s their algorithmic bytes reach
with the chip to themselves (inside the step they share it with the side stream's GEMMs)."""
import os, sys
sys.path.insert(0, 'aes-lac-2018_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from ds2hip import ops
def tm(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))
rows, feat = int(os.environ.get('ROWS', '4050')), 800
xa = torch.randn(rows, 800, device='cuda'); xb = torch.randn(rows, 800, device='cuda'); dy = torch.randn(rows, feat, device='cuda')
rm, rv = torch.zeros(feat, device='cuda'), torch.ones(feat, device='cuda')
g = torch.ones(feat, device='cuda'); b = torch.zeros(feat, device='cuda'); dg = torch.empty_like(g); db = torch.empty_like(g)
mi = ops.bn1d_stats(xa, xb, rows, feat, rm, rv, True)
n = rows * feat * 4
for name, f, nbytes in (('bn1d_stats (read xa, xb)', lambda: ops.bn1d_stats(xa, xb, rows, feat, rm, rv, True), 2 * n),
                        ('bn1d_apply (read xa, xb, write y)', lambda: ops.bn1d_apply(xa, xb, mi, g, b, rows, feat), 3 * n),
                        ('bn1d_bwd (reduce: xa, xb, dy; apply: xa, xb, dy -> dx)', lambda: ops.bn1d_bwd(xa, xb, dy, mi, g, rows, feat, dg, db), 7 * n)):
    t = tm(f)
    print('%-64s %7.1f us  %5.2f TB/s = %4.1f %% of 8 TB/s' % (name, t, nbytes / t / 1e6, 100 * nbytes / t / 1e6 / 8))
for c, d, t_ in ((32, 61, 415), (32, 21, 405)):
    x = torch.randn(10, c, d, t_, device='cuda'); dyc = torch.randn_like(x)
    rm2, rv2 = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda'); g2 = torch.ones(c, device='cuda'); b2 = torch.zeros(c, device='cuda')
    dg2, db2 = torch.empty_like(g2), torch.empty_like(g2)
    mi2 = ops.bn2d_stats(x, rm2, rv2, True)
    n2 = x.numel() * 4
    for name, f, nbytes in (('bn2d_stats %dx%d' % (d, t_), lambda: ops.bn2d_stats(x, rm2, rv2, True), n2),
                            ('bn2d_apply_htanh', lambda: ops.bn2d_apply_htanh(x, mi2, g2, b2, False), 2 * n2),
                            ('bn2d_htanh_bwd', lambda: ops.bn2d_htanh_bwd(x, dyc, mi2, g2, b2, dg2, db2), 5 * n2)):
        t = tm(f)
        print('%-64s %7.1f us  %5.2f TB/s = %4.1f %% of 8 TB/s' % (name, t, nbytes / t / 1e6, 100 * nbytes / t / 1e6 / 8))

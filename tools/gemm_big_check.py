"""The 256 x 128 one-wave-per-SIMD GEMM (gemm_bf16x_big_kernel) against the 128 x 128 kernel: results and time.
   DS2_LIB_VARIANT=tuning python tools/gemm_big_check.py [ROWS ...]      (spawns itself once per DS2_GEMM_BIG value)"""
import os, sys, subprocess, json
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(_ROOT, 'aes-lac-2018_amd')); sys.path.insert(0, _ROOT)

def worker(rows_list):
    import torch, numpy as np
    from ds2hip import ops
    out = {}
    for rows in rows_list:
        shapes = [('gi   NT', 0, 1, rows, 4800, 800), ('gi0  NT', 0, 1, rows, 4800, 672), ('dX   NN', 0, 0, rows, 800, 4800),
                  ('dWih TN', 1, 0, 4800, 800, rows), ('dWhh TN', 1, 0, 1600, 800, rows), ('sq   NT', 0, 1, 4096, 4096, 4096)]
        for name, ta, tb, m, n, k in shapes:
            g = torch.Generator(device='cuda'); g.manual_seed(m + n + k)
            a = torch.randn((k, m) if ta else (m, k), device='cuda', generator=g)
            b = torch.randn((n, k) if tb else (k, n), device='cuda', generator=g)
            c = torch.empty(m, n, device='cuda')
            for _ in range(3): ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=c, split_k=0)
            torch.cuda.synchronize()
            ts = []
            for _ in range(10):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=c, split_k=0); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            # fp64 reference on a sample of rows (the full product of the large shapes is slow in fp64)
            idx = torch.randint(0, m, (256,), device='cuda', generator=g)
            aa = (a.t() if ta else a)[idx].double(); bb = (b.t() if tb else b).double()
            ref = aa @ bb
            scale = (aa.abs() @ bb.abs())
            err = float(((c[idx].double() - ref).abs() / scale).max())
            out['%s %d' % (name, rows)] = (float(np.median(ts)), 2.0 * m * n * k / float(np.median(ts)) / 1e6, err)
    print(json.dumps(out))

if __name__ == '__main__':
    if os.environ.get('_GEMM_BIG_WORKER') == '1':
        worker([int(x) for x in sys.argv[1:]])
    else:
        rows = sys.argv[1:] or ['4050']
        res = {}
        for big in ('0', '2'):
            env = dict(os.environ, _GEMM_BIG_WORKER='1', DS2_GEMM_BIG=big, DS2_LIB_VARIANT='tuning')
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + rows, env=env, capture_output=True, text=True)
            if r.returncode != 0:
                print(r.stderr[-2000:]); sys.exit(1)
            res[big] = json.loads(r.stdout.strip().splitlines()[-1])
        for k in res['0']:
            a, b = res['0'][k], res['2'][k]
            print('%-16s 128x128: %8.1f us %6.1f TF err %.1e | 256x128: %8.1f us %6.1f TF err %.1e | %+5.1f %%'
                  % (k, a[0], a[1], a[2], b[0], b[1], b[2], 100.0 * (a[0] / b[0] - 1.0)))

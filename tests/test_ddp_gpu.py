"""The data-parallel step at world size 2, numerically: two fresh processes on cuda:0 (gloo on device tensors)
against an oracle that emulates DistributedDataParallel (tests/ddp_common.py).

Checked: the construction-time broadcast (rank 1 starts from other weights), per-replica BatchNorm statistics,
per-rank loss/B, the gradient MEAN over ranks (sum all-reduce of per-layer slices + the 1/W fold in the fused update),
the global-norm clip on the averaged gradient (max_norm small enough to engage), Nesterov momentum -- with the
all-reduce overlapped with backward on the side stream and, separately, issued after backward.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('overlap,gru_mode', [('1', 'step'), ('0', 'step'), ('1', 'persistent')])
def test_two_rank_step_matches_ddp_oracle(tmp_path, overlap, gru_mode):
    """``gru_mode='persistent'`` (round 5): the same two ranks with the PERSISTENT recurrence kernels -- the forms the product
    runs (gru_fwd_persistent4_kernel / gru_bwd_persistent4_kernel, speculative hand-off) -- so that the overlapped all-reduce
    (per-layer slices on the comm stream, gated by events of the main and the weight-gradient stream) meets persistent
    launches of BOTH ranks on one GPU.  At the test model's width (H = 32) a launch is 8-24 workgroups: the two ranks'
    launches are co-resident together, which two full-width ones (174-240 workgroups each) on ONE GPU could not be -- on a real
    node every rank has its own GPU.  DS2_GRU_STRICT=1: a fall-back to the per-step kernels fails the rank."""
    from tests import ddp_common as dc
    world = 2
    port = 29700 + (os.getpid() % 200) + (7 if overlap == '1' else 0) + (13 if gru_mode == 'persistent' else 0)
    env = dict(os.environ, DS2_GRU_MODE=gru_mode, DS2_ALLREDUCE_OVERLAP=overlap, DS2_GRU_STRICT='1')
    outs = [str(tmp_path / ('rank%d.npz' % r)) for r in range(world)]

    def run_ranks(port):
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'ddp_worker.py'), str(r), str(world),
                                   str(port), outs[r]], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                 for r in range(world)]
        logs = []
        for p in procs:
            try:
                so, se = p.communicate(timeout=300)      # (a hung rank dumps its stack and exits by itself after 200 s)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            logs.append(se[-3000:])
        return all(p.returncode == 0 for p in procs), logs

    ok, logs = run_ranks(port)
    # no retry: a rank that stalls (it dumps its stacks and exits after 200 s) fails the test with those stacks -- an
    # intermittent deadlock of the overlapped all-reduce (comm stream, side stream, gate events) must stay visible
    assert ok, '\n'.join(logs)
    want_losses, want_sd, want_params = dc.oracle_ddp_steps(world)
    got = [np.load(o) for o in outs]
    assert int(got[0]['overlap']) == int(overlap)
    for r in range(world):
        np.testing.assert_allclose(got[r]['losses'], want_losses[r], rtol=2e-4, err_msg='rank %d losses' % r)
        for i, w in enumerate(want_params[r]):
            np.testing.assert_allclose(got[r]['p%03d' % i], w, atol=5e-5, err_msg='rank %d param %d' % (r, i))
    # the replicas stay bit-identical to each other (same averaged gradient, same update)
    for i in range(len(want_params[0])):
        assert np.array_equal(got[0]['p%03d' % i], got[1]['p%03d' % i])
    assert np.array_equal(got[0]['norms'], got[1]['norms'])
    # rank 0's running statistics are its own batches' (per-replica BatchNorm)
    for k, v in want_sd.items():
        if 'running' in k:
            np.testing.assert_allclose(got[0]['buf_' + k], v.numpy(), rtol=1e-4, atol=1e-5, err_msg=k)

"""Process-environment settings of the data-parallel step (one process per GPU, RCCL over xGMI) -- ONE place, shared by
``train.py`` and ``bench.py``, so that what the benchmark measures is what training runs with.

Both knobs are read by the HIP runtime / RCCL when they start, so ``data_parallel_env()`` has to run BEFORE the first
``torch.cuda`` call of the process (``torch.cuda.is_available()`` already initialises the runtime).  Both are
``setdefault``: an operator's own value wins.

  GPU_MAX_HW_QUEUES = 3    with a process group the step uses four streams (main, weight-gradient, all-reduce, RCCL's
                           own).  Measured on one rank (bench.py DS2_BENCH_FORCE_DIST=1): 308 k frames/s with the
                           runtime's default four hardware queues, 350 k -- the rate without a process group -- with two
                           or three.
  NCCL_MAX_NCHANNELS = 32  a persistent backward-recurrence launch needs its ~204 workgroups co-resident, one per CU, and
                           leaves 52 CUs free; the gradient all-reduce of the layers above runs beside it.  RCCL's kernels
                           take one workgroup per channel: capped at 32 they always fit on the free CUs, so a collective
                           can delay a recurrence launch (the bounded spins cover that) but never keep one from becoming
                           resident.  The forward recurrence (240 CUs) never meets a collective: the step waits for its
                           all-reduce before the optimizer, i.e. before the next forward pass.

Neither value has been measured with more than one rank (no multi-GPU box was in reach of rounds 1-4): they are the
settings DESIGN.md section 5's predictions assume, and ``assert_no_fallbacks`` is the check that fails loudly when the
co-residency assumption behind them does not hold on a real node.
"""
import os

DEFAULTS = {'GPU_MAX_HW_QUEUES': '3', 'NCCL_MAX_NCHANNELS': '32'}


def data_parallel_env(environ=None):
    """Apply the defaults above to ``environ`` (default: this process) and return the values now in force."""
    env = os.environ if environ is None else environ
    for k, v in DEFAULTS.items():
        env.setdefault(k, v)
    return {k: env[k] for k in DEFAULTS}


def assert_no_fallbacks(count, where='this run'):
    """A persistent recurrence launch that could not run (grid not co-resident, hand-off time-out) switches the device to
    the launch-per-step kernels, 3x slower.  ``train.py`` logs that and goes on -- a training run should survive; a
    benchmark or a scaling test must FAIL instead of reporting the fall-back's rate."""
    if int(count) > 0:
        raise RuntimeError('%d persistent recurrence launch(es) fell back to the launch-per-step kernels in %s: the '
                           'co-residency settings (%s) do not hold on this node -- see codes/utils/dist_utils.py'
                           % (int(count), where, ', '.join('%s=%s' % (k, os.environ.get(k)) for k in DEFAULTS)))

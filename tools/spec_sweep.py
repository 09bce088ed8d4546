"""Sweep the speculative hand-off's first-attempt delay / re-load backoff (DS2_GRU_SPEC_FWD / DS2_GRU_SPEC_BWD, read per launch)."""
# (the knobs this tool sweeps are TUNING knobs: read only by `python aes-lac-2018_amd/csrc/build.py --variant tuning`,
# i.e. run it with DS2_LIB_VARIANT=tuning -- the release library ignores them; csrc/ds2_common.h: ds2_tune_env)
import os, sys
os.environ.setdefault('DS2_SWEEP_LIB', '')
sys.argv = sys.argv[:1]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gru_sweep as g
for bsz in [int(v) for v in os.environ.get('SIZES', '10,8,4').split(',')]:
    for delay in [int(v) for v in os.environ.get('DELAYS', '4,8,12,16,20,24,28').split(',')]:
        for back in [int(v) for v in os.environ.get('BACKS', '2,6').split(',')]:
            os.environ['DS2_GRU_SPEC_FWD'] = os.environ['DS2_GRU_SPEC_BWD'] = '%d,%d' % (delay, back)
            f, b = g.measure(bsz, reps=4)
            print('B=%2d delay %2d backoff %d  fwd %.2f  bwd %.2f us/step' % (bsz, delay, back, f, b), flush=True)

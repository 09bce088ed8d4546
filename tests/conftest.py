import os
import sys

# The CPU oracle (torch on OpenMP) runs with one thread per CPU, and libgomp's threads SPIN at every barrier: on a box whose
# CPUs are shared with anything else (another test process, a noisy neighbour of this VM) a spinning thread holds the CPU the
# thread it waits for needs -- measured here, two copies of one oracle test side by side: 295 s each spinning, 29 s each
# sleeping; the whole CPU suite varied between 47 s and 13 min from run to run.  Must be set before libgomp is loaded, i.e.
# before anything imports torch (this file is the first thing pytest imports from the tree).
os.environ.setdefault('OMP_WAIT_POLICY', 'PASSIVE')
os.environ.setdefault('GOMP_SPINCOUNT', '0')

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'aes-lac-2018_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN

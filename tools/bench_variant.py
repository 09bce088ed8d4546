"""bench.py on a variant build of the library (same-box A/B runs): DS2_LIB_VARIANT=<name> python tools/bench_variant.py <bench.py arguments>
loads aes-lac-2018_amd/ds2hip/libds2hip_<name>.so (csrc/build.py: build_variant / build_gemm_variant) instead of libds2hip.so."""
import os, runpy, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'aes-lac-2018_amd')); sys.path.insert(0, ROOT)
from ds2hip import lib
if os.environ.get('DS2_LIB_VARIANT'):
    lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), 'libds2hip_%s.so' % os.environ['DS2_LIB_VARIANT'])
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name='__main__')

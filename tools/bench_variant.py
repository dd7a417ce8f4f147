"""Development aid: `python tools/bench_variant.py LIB.so [bench.py flags]` runs bench.py against another
build of the library (kernel experiments: several variants in one GPU session)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nadavca_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ['bench.py'] + sys.argv[2:]
import bench
bench.main()

"""tools/bench_engine.py against another build of the library: python tools/ab_bench_engine.py <path to .so> <bench_engine args>."""
import sys
sys.path.insert(0, ".")
import jukebox_amd._lib as L
L.LIB_PATH = sys.argv[1]
sys.argv = [sys.argv[0]] + sys.argv[2:]
from tools import bench_engine
bench_engine.main()

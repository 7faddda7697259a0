import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fusion_amd import _lib
for p in sys.argv[1:]:
    _lib._lib = None; _lib.LIB_PATH = os.path.abspath(p)
    print(os.path.basename(p), flush=True)
    from tools.bench_kernels import bench_maxsim
    bench_maxsim(Qs=(195, 1024))

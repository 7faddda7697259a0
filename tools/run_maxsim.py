import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_kernels import bench_maxsim
bench_maxsim(Qs=(195,))

"""What a radix pass and the fixed parts of the row sort cost: keys with 0..4 varying bytes (fp32) and fp64 variants -- the DIGIT passes
(the bucket ranking of fp32 rows is switched off here; tools/run_sort_ab.py compares the two).  Usage: python tools/sort_pass_cost.py [lib.so ...]"""
import os, sys, torch
os.environ["FZ_SORT_BUCKET_RANK"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fusion_amd import _lib, ops
from tools.bench_sort import timeit
Q, N = 1024, 27942

def run(tag, only64=False):
    g = torch.Generator(device="cuda").manual_seed(0)
    def plane(dt, f):
        k = ops.alloc_plane(Q, N, dt, "cuda"); k.copy_(f()); return k
    r = lambda dt=torch.float32: torch.rand((Q, N), generator=g, device="cuda", dtype=dt)
    cases = {} if only64 else {
        "f32 [-1,1) 4 passes": plane(torch.float32, lambda: r() * 2 - 1),
        "f32 [1,2) 3 passes": plane(torch.float32, lambda: r() + 1),
        "f32 const 0 passes": plane(torch.float32, lambda: torch.ones((Q, N), device="cuda")),
    }
    cases.update({
        "f64 [0,1)": plane(torch.float64, lambda: r(torch.float64)),
        "f64 [1,2)": plane(torch.float64, lambda: r(torch.float64) + 1),
        "f64 const": plane(torch.float64, lambda: torch.ones((Q, N), device="cuda", dtype=torch.float64)),
        "f64 BM25-like (40 % zeros)": plane(torch.float64, lambda: (torch.distributions.Gamma(0.5, 0.25).sample((Q, N)).to("cuda").double() - 2.0).clamp_min(0.0)),
    })
    out = {k: round(timeit(lambda: ops.sort_rows_desc(v, want_keys=False, want_rank=True)), 4) for k, v in cases.items()}
    print(tag, out, flush=True)

if __name__ == "__main__":
    libs = sys.argv[1:] or [_lib.LIB_PATH]
    for p in libs:
        _lib._lib = None
        _lib.LIB_PATH = os.path.abspath(p)
        run(os.path.basename(p), only64="only64" in p)

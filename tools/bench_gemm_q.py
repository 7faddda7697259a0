import sys, json, torch
sys.path.insert(0, ".")
from fusion_amd import ops
from tools.bench_kernels import timeit
g = torch.Generator(device="cuda").manual_seed(1)
N = 27942
for d in (768, 32008):
    Dn = ops.normalize_rows(torch.randn((N, d), generator=g, device="cuda"))
    for Q in (1024, 224, 208, 195, 193, 192, 201, 200, 160, 128, 64):
        Qn = ops.normalize_rows(torch.randn((Q, d), generator=g, device="cuda"))
        out = ops.alloc_plane(Q, N, torch.float32, "cuda")
        ms = timeit(lambda: ops.dot_scores(Qn, Dn, out=out), n=20 if d == 768 else 3, warm=2)
        print(json.dumps(dict(Q=Q, d=d, ms=round(ms, 4), tflops=round(2.0 * Q * N * d / ms / 1e9, 1), frac=round(2.0 * Q * N * d / ms / 1e9 / 157.3, 3))), flush=True)
    del Dn

import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fusion_amd import ops
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
g = torch.Generator(device="cuda").manual_seed(0)
Qn = ops.normalize_rows(torch.randn((1024, 768), generator=g, device="cuda"))
Dall = ops.normalize_rows(torch.randn((40000, 768), generator=g, device="cuda"))
for N in (4096, 8192, 12288, 16384, 20480, 24576, 26624, 27942, 28672, 30720, 32768):
    Dn = Dall[:N]
    out = ops.alloc_plane(1024, N, torch.float32, "cuda")
    ms = timeit(lambda: ops.dot_scores(Qn, Dn, out=out))
    blocks = 8 * ((N + 127) // 128 + 7) // 8 * 8
    print(f"N={N:6d} tiles={8*((N+127)//128):5d} rounds={8*((N+127)//128)/512:5.2f} ms={ms:.4f} TF={2*1024*N*768/ms/1e9:.1f}", flush=True)

"""Score GEMM against the vendor fp32 GEMM on the same operands, random and all-zero data (clock/power sensitivity)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fusion_amd import ops


def timeit(f, n=10, warm=3):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    d = 768
    for Q, N in ((1024, 27942), (1024, 276307), (195, 27942), (4096, 65536)):
        g = torch.Generator(device="cuda").manual_seed(1)
        for kind in ("randn", "zeros"):
            if kind == "randn":
                Qn = ops.normalize_rows(torch.randn((Q, d), generator=g, device="cuda"))
                Dn = ops.normalize_rows(torch.randn((N, d), generator=g, device="cuda"))
            else:
                Qn = torch.zeros((Q, d), device="cuda"); Dn = torch.zeros((N, d), device="cuda")
            out = ops.alloc_plane(Q, N, torch.float32, "cuda")
            ms = timeit(lambda: ops.dot_scores(Qn, Dn, out=out))
            ref = torch.empty((Q, N), device="cuda")
            DnT = Dn.t()
            ms_v = timeit(lambda: torch.mm(Qn, DnT, out=ref))
            fl = 2.0 * Q * N * d / 1e9
            print(f"Q={Q} N={N} {kind}: ours {ms:.3f} ms {fl/ms:.1f} TF/s | vendor {ms_v:.3f} ms {fl/ms_v:.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()

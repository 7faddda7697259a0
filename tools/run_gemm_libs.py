"""Time the score GEMM of several builds of the library (ablation builds, see tools/ablate/README.md)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fusion_amd import _lib, ops
from tools.diag_gemm import timeit

Q, N, d = 1024, 276307, 768
g = torch.Generator(device="cuda").manual_seed(1)
Qn = ops.normalize_rows(torch.randn((Q, d), generator=g, device="cuda"))
Dn = ops.normalize_rows(torch.randn((N, d), generator=g, device="cuda"))
out = ops.alloc_plane(Q, N, torch.float32, "cuda")
Qz, Dz = torch.zeros_like(Qn), torch.zeros_like(Dn)
for p in sys.argv[1:]:
    _lib._lib = None; _lib.LIB_PATH = os.path.abspath(p)
    ref = Qn[:128].double() @ Dn[:4096].double().t()
    err = (ops.dot_scores(Qn, Dn, out=out)[:128, :4096].double() - ref).abs().max().item()
    ms = timeit(lambda: ops.dot_scores(Qn, Dn, out=out), n=20)
    mz = timeit(lambda: ops.dot_scores(Qz, Dz, out=out), n=20)
    print(f"{os.path.basename(p)}: randn {ms:.3f} ms {2.0 * Q * N * d / ms / 1e9:.1f} TF/s | zeros {mz:.3f} ms {2.0 * Q * N * d / mz / 1e9:.1f} TF/s | max_err {err:.2e}", flush=True)
